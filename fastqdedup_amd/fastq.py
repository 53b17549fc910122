"""FASTQ in, keys out, kept records back out -- the glue on either side of the hot
path. The reference does this with dnaio/xopen
(`src/fastqdedup/__init__.py:54-57, 160-206`), one Python object per record; here a
file becomes one uint8 buffer plus numpy index arrays, so 50 M records stay 50 M
rows, not 50 M objects.

Reproduced behaviour (reference file:line):
  * records of several files are zipped and stop at the shortest file (:180);
  * mates must carry the same id (:181-185) -- ids compared up to the first blank,
    ignoring a trailing 1/2/3, as dnaio's ``records_are_mates`` does;
  * ``--check-lengths`` slices use Python slice semantics on the sequence AND on the
    quality string of each file (:160-167, :243-251);
  * pass 2 writes ``@name\\nseq\\n+\\nqual\\n`` (dnaio ``fastq_bytes``), gzip level 1 for
    ``.gz`` names (:197-198).
"""
from __future__ import annotations

import bz2
import gzip
import lzma
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np


class FastqFormatError(Exception):
    """Same role as dnaio.FastqFormatError."""

    def __init__(self, msg: str, line: Optional[int] = None):
        where = "unknown line" if line is None else f"line {line + 1}"
        super().__init__(f"Error in FASTQ file at {where}: {msg}")


def _open_read(path: str):
    with open(path, "rb") as fh:
        magic = fh.read(6)
    if magic[:2] == b"\x1f\x8b":
        return gzip.open(path, "rb")
    if magic[:3] == b"BZh":
        return bz2.open(path, "rb")
    if magic[:6] == b"\xfd7zXZ\x00":
        return lzma.open(path, "rb")
    return open(path, "rb")


def open_write(path: str):
    """xopen(mode="wb", compresslevel=1, threads=0) by file extension."""
    if path.endswith(".gz"):
        return gzip.open(path, "wb", compresslevel=1)
    if path.endswith(".bz2"):
        return bz2.open(path, "wb", compresslevel=1)
    if path.endswith(".xz"):
        return lzma.open(path, "wb", preset=1)
    return open(path, "wb")


@dataclass
class FastqTable:
    """One FASTQ file as a byte buffer + per-record spans (half-open byte ranges)."""
    buf: np.ndarray            # uint8, the whole decompressed file
    rec_start: np.ndarray      # int64[n]   '@'
    rec_end: np.ndarray        # int64[n]   one past the record's last newline (or EOF)
    name_start: np.ndarray     # after '@'
    name_end: np.ndarray
    seq_start: np.ndarray
    seq_end: np.ndarray
    qual_start: np.ndarray
    qual_end: np.ndarray
    normalized: bool           # every record is exactly "@name\\nseq\\n+\\nqual\\n"

    def __len__(self) -> int:
        return len(self.rec_start)

    def head(self, n: int) -> "FastqTable":
        return FastqTable(self.buf, *(a[:n] for a in (
            self.rec_start, self.rec_end, self.name_start, self.name_end, self.seq_start, self.seq_end,
            self.qual_start, self.qual_end)), self.normalized)


def parse_fastq(data, first_line: int = 0) -> FastqTable:
    """Whole records in ``data`` (bytes or uint8 array) -> table. ``first_line``: the line number of
    the buffer's first line in its file, for error messages."""
    buf = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
    nl = np.flatnonzero(buf == 10).astype(np.int64)
    ends = nl
    if len(buf) and (len(nl) == 0 or nl[-1] != len(buf) - 1):
        ends = np.append(nl, len(buf))                     # last line without a newline
    starts = np.concatenate([[0], ends[:-1] + 1]).astype(np.int64) if len(ends) else np.zeros(0, np.int64)
    n_lines = len(ends)
    while n_lines and starts[n_lines - 1] >= ends[n_lines - 1] and n_lines % 4:   # trailing blank lines
        n_lines -= 1
    if n_lines % 4:
        raise FastqFormatError("Premature end of file encountered.", line=first_line + n_lines)
    starts, ends = starts[:n_lines], ends[:n_lines]
    # '\r\n' line ends: drop the '\r'
    content_end = ends.copy()
    has_cr = (content_end > starts) & (buf[np.maximum(content_end - 1, 0)] == 13)
    content_end[has_cr] -= 1
    l0, l1, l2, l3 = (slice(k, None, 4) for k in range(4))
    if n_lines:
        bad = np.flatnonzero((buf[starts[l0]] != ord("@")) | (content_end[l0] == starts[l0]))
        if len(bad):
            raise FastqFormatError("Line expected to start with '@'", line=first_line + int(bad[0]) * 4)
        plus_ok = (content_end[l2] > starts[l2]) & (buf[np.minimum(starts[l2], len(buf) - 1)] == ord("+"))
        bad = np.flatnonzero(~plus_ok)
        if len(bad):
            raise FastqFormatError("Line expected to start with '+'", line=first_line + int(bad[0]) * 4 + 2)
        bad = np.flatnonzero((content_end[l1] - starts[l1]) != (content_end[l3] - starts[l3]))
        if len(bad):
            raise FastqFormatError("Length of sequence and qualities differ", line=first_line + int(bad[0]) * 4 + 3)
    rec_end = np.minimum(ends[l3] + 1, len(buf)) if n_lines else np.zeros(0, np.int64)
    normalized = bool(n_lines == 0 or (not has_cr.any() and np.all(content_end[l2] - starts[l2] == 1)
                                       and np.all(buf[np.minimum(ends[l3], len(buf) - 1)] == 10)
                                       and int(ends[-1]) < len(buf)))
    return FastqTable(buf, starts[l0], rec_end, starts[l0] + 1, content_end[l0], starts[l1], content_end[l1],
                      starts[l3], content_end[l3], normalized)


def read_fastq(path: str) -> FastqTable:
    with _open_read(path) as fh:
        return parse_fastq(fh.read())


class FastqChunks:
    """A FASTQ file as a sequence of tables of at most ``chunk_records`` records each: the file is
    decompressed block by block and never held whole (the reference streams records one by one,
    __init__.py:54-57; a 50 M-pair run is > 30 GB of text)."""

    def __init__(self, path: str, chunk_records: int, block_bytes: int = 64 << 20):
        self.path, self.chunk_records, self.block_bytes = path, int(chunk_records), int(block_bytes)

    def __iter__(self):
        want_lines = 4 * self.chunk_records
        with _open_read(self.path) as fh:
            pending: List[bytes] = []       # blocks of the chunk being assembled
            pending_lines = 0
            line0 = 0
            eof = False
            while not eof or pending:
                block = b"" if eof else fh.read(self.block_bytes)
                if not block:
                    eof = True
                    if pending:
                        data = b"".join(pending)
                        pending, pending_lines = [], 0
                        table = parse_fastq(data, line0)
                        if len(table):
                            yield table
                    break
                arr = np.frombuffer(block, dtype=np.uint8)
                nl = np.flatnonzero(arr == 10)
                start = 0
                while pending_lines + (len(nl) - np.searchsorted(nl, start)) >= want_lines:
                    # the chunk's last newline lies in this block
                    first_nl = int(np.searchsorted(nl, start))
                    cut = int(nl[first_nl + (want_lines - pending_lines) - 1]) + 1
                    data = b"".join(pending + [block[start:cut]])
                    pending, pending_lines = [], 0
                    yield parse_fastq(data, line0)
                    line0 += want_lines
                    start = cut
                if start < len(block):
                    pending.append(block[start:])
                    pending_lines += len(nl) - int(np.searchsorted(nl, start))


def zip_chunks(paths: Sequence[str], chunk_records: int):
    """Chunk k of every file, cut to the shortest: ``(tables, n, first record number)``; stops with
    the shortest file like the reference's zip() over its readers (__init__.py:180)."""
    its = [iter(FastqChunks(p, chunk_records)) for p in paths]
    base = 0
    while True:
        tables = []
        for it in its:
            t = next(it, None)
            if t is None:
                return
            tables.append(t)
        n = min(len(t) for t in tables)
        if n:
            yield tables, n, base
        base += n
        if any(len(t) < chunk_records for t in tables):
            return          # some file has ended: a tuple needs a record of every file


# ---------------------------------------------------------------------------
# mates
# ---------------------------------------------------------------------------

def _id_ends(t: FastqTable) -> np.ndarray:
    """End of the record id: first space or tab of the name (or the end of the name)."""
    blank = np.flatnonzero((t.buf == 32) | (t.buf == 9)).astype(np.int64)
    if not len(blank):
        return t.name_end.copy()
    pos = np.searchsorted(blank, t.name_start)
    nxt = blank[np.minimum(pos, len(blank) - 1)]
    hit = (pos < len(blank)) & (nxt < t.name_end)
    return np.where(hit, nxt, t.name_end)


def check_mates(tables: Sequence[FastqTable], n: int) -> None:
    """dnaio.records_are_mates over the first n records of every file (reference :181-185)."""
    if len(tables) < 2 or n == 0:
        return
    first = tables[0]
    e0 = _id_ends(first)[:n]
    s0 = first.name_start[:n]
    len0 = e0 - s0
    last0 = first.buf[np.maximum(e0 - 1, 0)]
    digit0 = (len0 > 0) & (last0 >= ord("1")) & (last0 <= ord("3"))
    for t in tables[1:]:
        e = _id_ends(t)[:n]
        s = t.name_start[:n]
        ln = e - s
        ok = ln == len0
        last = t.buf[np.maximum(e - 1, 0)]
        strip = digit0 & (ln > 0) & (last >= ord("1")) & (last <= ord("3"))
        cmp_len = np.where(strip, len0 - 1, len0)
        # compare id bytes: a ragged compare, done per distinct length (few in practice)
        for L in np.unique(cmp_len[ok]):
            rows = np.flatnonzero(ok & (cmp_len == L))
            if L == 0 or not len(rows):
                continue
            a = first.buf[s0[rows, None] + np.arange(L)]
            b = t.buf[s[rows, None] + np.arange(L)]
            ok[rows[np.any(a != b, axis=1)]] = False
        bad = np.flatnonzero(~ok)
        if len(bad):
            i = int(bad[0])
            names = ", ".join(bytes(x.buf[x.name_start[i]:x.name_end[i]]).decode("ascii", "replace")
                              for x in tables)
            raise FastqFormatError(f"FASTQ files not in sync: {names} are not mates.", line=None)


# ---------------------------------------------------------------------------
# keys
# ---------------------------------------------------------------------------

def _slice_bounds(slc: Optional[slice], lens: np.ndarray):
    """Vectorised ``slice.indices`` for step 1: (start, length) per record."""
    if slc is None:
        return np.zeros_like(lens), lens.copy()
    step = 1 if slc.step is None else slc.step
    if step != 1:
        return None

    def norm(v, default):
        if v is None:
            return np.full_like(lens, default) if np.isscalar(default) else default.copy()
        v = int(v)
        out = np.full_like(lens, v)
        if v < 0:
            out = np.maximum(lens + v, 0)
        return np.minimum(out, lens)
    start = norm(slc.start, 0)
    stop = norm(slc.stop, lens)
    return start, np.maximum(stop - start, 0)


def _ragged_gather(buf: np.ndarray, seg_start: np.ndarray, seg_len: np.ndarray, out: np.ndarray,
                   out_pos: np.ndarray, chunk: int = 2_000_000) -> None:
    """out[out_pos[i] : out_pos[i] + seg_len[i]] = buf[seg_start[i] : seg_start[i] + seg_len[i]]."""
    n = len(seg_start)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        ln = seg_len[a:b]
        total = int(ln.sum())
        if not total:
            continue
        excl = np.cumsum(ln) - ln
        within = np.arange(total, dtype=np.int64) - np.repeat(excl, ln)
        out[np.repeat(out_pos[a:b], ln) + within] = buf[np.repeat(seg_start[a:b], ln) + within]


def build_strings(tables: Sequence[FastqTable], slices: Optional[Sequence[Optional[slice]]], n: int,
                  what: str) -> Tuple[np.ndarray, np.ndarray, int]:
    """``joinfunc(record.<what> for record in tuple)`` for the first n tuples (reference :160-167,
    :243-251): (bytes, offsets[n+1], fixed length or 0)."""
    parts = []
    for f, t in enumerate(tables):
        start = (t.seq_start if what == "sequence" else t.qual_start)[:n]
        end = (t.seq_end if what == "sequence" else t.qual_end)[:n]
        lens = end - start
        slc = None if slices is None else slices[f]
        fast = _slice_bounds(slc, lens)
        if fast is not None:
            s, ln = fast
            parts.append((t.buf, start + s, ln, None))
        else:                       # stepped slice: materialise per distinct length
            idx_rows = []
            ln = np.zeros(n, dtype=np.int64)
            for L in np.unique(lens):
                rows = np.flatnonzero(lens == L)
                sel = np.arange(int(L))[slc]
                ln[rows] = len(sel)
                idx_rows.append((rows, sel))
            parts.append((t.buf, start, ln, idx_rows))
    total_len = np.zeros(n, dtype=np.int64)
    for _, _, ln, _ in parts:
        total_len += ln
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(total_len)
    out = np.empty(int(offsets[n]), dtype=np.uint8)
    pos = offsets[:-1].astype(np.int64)
    for buf, start, ln, idx_rows in parts:
        if idx_rows is None:
            _ragged_gather(buf, start, ln, out, pos)
        else:
            for rows, sel in idx_rows:
                if len(sel):
                    out[pos[rows, None] + np.arange(len(sel))] = buf[start[rows, None] + sel]
        pos = pos + ln
    fixed = int(total_len[0]) if n and np.all(total_len == total_len[0]) else 0
    return out, offsets, fixed


# ---------------------------------------------------------------------------
# pass 2
# ---------------------------------------------------------------------------

def write_records_to(table: FastqTable, keep: np.ndarray, out) -> None:
    """Append records `keep` (ascending record numbers of this table) as dnaio's ``fastq_bytes`` would."""
    keep = np.asarray(keep, dtype=np.int64)
    if not len(keep):
        return
    if table.normalized:
        mark = np.zeros(len(table.buf) + 1, dtype=np.int8)
        np.add.at(mark, table.rec_start[keep], 1)
        np.add.at(mark, table.rec_end[keep], -1)
        out.write(table.buf[np.cumsum(mark[:-1]) > 0].tobytes())
        return
    b = table.buf
    for i in keep:
        out.write(b"@" + b[table.name_start[i]:table.name_end[i]].tobytes() + b"\n" +
                  b[table.seq_start[i]:table.seq_end[i]].tobytes() + b"\n+\n" +
                  b[table.qual_start[i]:table.qual_end[i]].tobytes() + b"\n")


def write_records(table: FastqTable, keep: np.ndarray, path: str) -> None:
    """Write records `keep` (ascending record numbers) to a new file."""
    with open_write(path) as out:
        write_records_to(table, keep, out)


def read_all(paths: List[str]) -> Tuple[List[FastqTable], int]:
    tables = [read_fastq(p) for p in paths]
    n = min(len(t) for t in tables) if tables else 0   # zip() stops at the shortest (reference :180)
    return tables, n
