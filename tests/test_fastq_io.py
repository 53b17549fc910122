"""Host-side glue of the CLI (fastqdedup_amd/fastq.py, cli.py parsing): FASTQ reader,
mate check, slice-based key builder, pass-2 writer. CPU only, no GPU needed."""
import gzip
import random

import numpy as np
import pytest

from fastqdedup_amd import fastq
from fastqdedup_amd.cli import argument_parser, length_string_to_slices


@pytest.mark.parametrize(["string", "result"], [          # reference tests/test_fastqdedup.py:27-34
    ("5,6,7", [slice(5), slice(6), slice(7)]),
    ("5:8,3,-5:3:-1", [slice(5, 8), slice(3), slice(-5, 3, -1)]),
    ("None:None:16", [slice(None, None, 16)]),
    ("::16", [slice(None, None, 16)]),
])
def test_length_string_to_slices(string, result):
    assert length_string_to_slices(string) == result


def _records(rng, n, lens=(5, 40)):
    out = []
    for i in range(n):
        L = rng.randint(*lens)
        out.append((f"read{i} extra/{i % 3}", "".join(rng.choice("ACGTN") for _ in range(L)),
                    "".join(chr(rng.randint(33, 74)) for _ in range(L))))
    return out


def _write(path, recs, eol="\n", plus_name=False, final_newline=True, gz=False):
    text = "".join(f"@{n}{eol}{s}{eol}+{n if plus_name else ''}{eol}{q}{eol}" for n, s, q in recs)
    if not final_newline:
        text = text[:-len(eol)]
    data = text.encode()
    (gzip.open if gz else open)(path, "wb").write(data)


@pytest.mark.parametrize("variant", ["plain", "gz", "crlf", "plusname", "nofinal"])
def test_reader_variants_and_writer_roundtrip(tmp_path, variant):
    rng = random.Random(1)
    recs = _records(rng, 300)
    p = str(tmp_path / ("in.fastq.gz" if variant == "gz" else "in.fastq"))
    _write(p, recs, eol="\r\n" if variant == "crlf" else "\n", plus_name=variant == "plusname",
           final_newline=variant != "nofinal", gz=variant == "gz")
    t = fastq.read_fastq(p)
    assert len(t) == len(recs)
    assert t.normalized == (variant in ("plain", "gz"))
    for i in (0, 1, 150, 299):
        assert bytes(t.buf[t.name_start[i]:t.name_end[i]]).decode() == recs[i][0]
        assert bytes(t.buf[t.seq_start[i]:t.seq_end[i]]).decode() == recs[i][1]
        assert bytes(t.buf[t.qual_start[i]:t.qual_end[i]]).decode() == recs[i][2]
    keep = np.array(sorted(rng.sample(range(300), 77)))
    for out in ("out.fastq", "out.fastq.gz"):
        o = str(tmp_path / out)
        fastq.write_records(t, keep, o)
        data = (gzip.open if out.endswith(".gz") else open)(o, "rb").read().decode()
        assert data == "".join(f"@{recs[i][0]}\n{recs[i][1]}\n+\n{recs[i][2]}\n" for i in keep)


def test_reader_rejects_malformed(tmp_path):
    p = str(tmp_path / "bad.fastq")
    open(p, "w").write("@a\nACGT\n+\nIII\n")
    with pytest.raises(fastq.FastqFormatError):
        fastq.read_fastq(p)
    open(p, "w").write("@a\nACGT\n+\nIIII\n@b\nAC\n")
    with pytest.raises(fastq.FastqFormatError):
        fastq.read_fastq(p)
    open(p, "w").write("a\nACGT\n+\nIIII\n")
    with pytest.raises(fastq.FastqFormatError):
        fastq.read_fastq(p)


@pytest.mark.parametrize("spec", ["16,16", "8", "4:12,::2", "-6:,3:-3", "None:None:-1,5", "100,0", "2:2,7:3"])
def test_key_builder_has_python_slice_semantics(tmp_path, spec):
    rng = random.Random(7)
    slices = length_string_to_slices(spec)
    files, all_recs = [], []
    n = 500
    for f in range(len(slices)):
        recs = _records(rng, n if f == 0 else n + 13)       # zip stops at the shortest file
        p = str(tmp_path / f"r{f}.fastq")
        _write(p, recs)
        files.append(p)
        all_recs.append(recs)
    tables, m = fastq.read_all(files)
    assert m == n
    for what, col in (("sequence", 1), ("qualities", 2)):
        raw, off, fixed = fastq.build_strings(tables, slices, m, what)
        want = ["".join(all_recs[f][i][col][slices[f]] for f in range(len(slices))) for i in range(m)]
        got = [bytes(raw[int(off[i]):int(off[i + 1])]).decode() for i in range(m)]
        assert got == want
        assert fixed == (len(want[0]) if len({len(w) for w in want}) == 1 else 0)
    raw, off, _ = fastq.build_strings(tables, None, m, "sequence")
    assert bytes(raw[int(off[5]):int(off[6])]).decode() == "".join(all_recs[f][5][1] for f in range(len(slices)))


def test_mates(tmp_path):
    def table(names):
        p = str(tmp_path / f"m{random.random()}.fastq")
        _write(p, [(nm, "ACGT", "IIII") for nm in names])
        return fastq.read_fastq(p)
    a = table(["x/1 first", "y.1", "same", "q1\tcomment", "tail2"])
    b = table(["x/2 second", "y.2 c", "same other", "q1", "tail3"])
    fastq.check_mates([a, b], 5)
    for bad in (["x/2", "y.2", "samE", "q1", "tail3"], ["x/2", "y.2", "same", "q", "tail3"],
                ["x/4", "y.2", "same", "q1", "tail3"], ["x/2", "yy.2", "same", "q1", "tail3"]):
        with pytest.raises(fastq.FastqFormatError, match="not mates"):
            fastq.check_mates([a, table(bad)], 5)


def test_argument_parser_surface():
    """Flags and defaults of the reference CLI (README.rst:42-89, __init__.py:305-361)."""
    a = argument_parser().parse_args(["r1.fq", "r2.fq"])
    assert (a.fastq, a.check_lengths, a.output, a.prefix) == (["r1.fq", "r2.fq"], None, None, "fastqdedup_R")
    assert (a.max_distance, a.max_average_error_rate, a.edit) == (1, 0.001, False)
    assert (a.cluster_dissection_method, a.verbose, a.quiet) == ("directional", 0, 0)
    b = argument_parser().parse_args(["-l", "16,8", "-o", "a", "-o", "b", "-p", "P", "-d", "2", "-E", "--edit",
                                      "-c", "adjacency", "-vv", "-q", "r1", "r2"])
    assert (b.check_lengths, b.output, b.prefix, b.max_distance) == ("16,8", ["a", "b"], "P", 2)
    assert (b.max_average_error_rate, b.edit, b.cluster_dissection_method) == (1.0, True, "adjacency")
    assert (b.verbose, b.quiet) == (2, 1)
    assert argument_parser().parse_args(["-e", "0.05", "x"]).max_average_error_rate == 0.05


@pytest.mark.parametrize("variant", ["plain", "gz", "crlf", "nofinal"])
def test_chunked_reader_equals_whole_file_reader(tmp_path, variant):
    """FastqChunks cuts a file into tables of whole records, whatever the block size (record ends
    falling on and off block borders, a last line without newline, CR LF)."""
    rng = random.Random(2)
    recs = _records(rng, 1003)
    p = str(tmp_path / ("in.fastq.gz" if variant == "gz" else "in.fastq"))
    _write(p, recs, eol="\r\n" if variant == "crlf" else "\n", final_newline=variant != "nofinal", gz=variant == "gz")
    for chunk, block in ((100, 257), (333, 1 << 16), (5000, 64), (1, 4096)):
        got = []
        for t in fastq.FastqChunks(p, chunk, block):
            assert 0 < len(t) <= chunk
            got += [(bytes(t.buf[a:b]).decode(), bytes(t.buf[c:d]).decode(), bytes(t.buf[e:f]).decode())
                    for a, b, c, d, e, f in zip(t.name_start, t.name_end, t.seq_start, t.seq_end, t.qual_start,
                                                t.qual_end)]
        assert got == recs, (chunk, block)
    with open(str(tmp_path / "broken.fastq"), "w") as fh:
        fh.write("@a\nACGT\n+\nIIII\n@b\nAC\n")
    with pytest.raises(fastq.FastqFormatError, match="Premature end of file"):
        list(fastq.FastqChunks(str(tmp_path / "broken.fastq"), 10))


def test_zip_chunks_stops_with_the_shortest_file(tmp_path):
    """reference __init__.py:180: zip() over the readers -- tuples end where the shortest file ends."""
    rng = random.Random(3)
    a, b = _records(rng, 950), _records(rng, 700)
    pa, pb = str(tmp_path / "a.fastq"), str(tmp_path / "b.fastq.gz")
    _write(pa, a)
    _write(pb, b, gz=True)
    seen, numbers = 0, []
    for tables, n, first in fastq.zip_chunks([pa, pb], 256):
        assert first == seen and all(len(t) >= n for t in tables)
        numbers.append(n)
        seen += n
    assert seen == 700 and numbers == [256, 256, 188]
