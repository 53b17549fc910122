"""The oracle's average_error_rate (restating reference _fastqmodule.c:38-76) against the fixture the
reference's own ``_fastq`` extension minted (tests/golden/fastq_error_rates.json: all 94 valid phred
characters, random strings, other offsets, error messages) and, where oracle/_ref is built, live
against that extension. Bit-exact (doubles compared by their hex form). CPU only."""
import json
import math
import os
import random

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fixture():
    with open(os.path.join(HERE, "golden", "fastq_error_rates.json")) as fh:
        return json.load(fh)


def _same(value: float, want: str) -> bool:
    return math.isnan(value) if want == "nan" else value.hex() == want


def test_every_valid_phred_character(oracle, fixture):
    assert len(fixture["single"]) == 94
    for ch, want in fixture["single"].items():
        assert _same(oracle.average_error_rate(ch), want), ch
        # the reference's table is 10 ** -(q / 10) (score_to_error_rate.py)
        assert float.fromhex(want) == 10 ** -((ord(ch) - 33) / 10)


def test_strings_and_offsets(oracle, fixture):
    for row in fixture["strings"]:
        got = oracle.average_error_rate(row["phred"], phred_offset=row["offset"])
        assert _same(got, row["value"]), row


def test_error_messages(oracle, fixture):
    for row in fixture["errors"]:
        with pytest.raises(ValueError) as err:
            oracle.average_error_rate(row["phred"], phred_offset=row["offset"])
        assert str(err.value) == row["message"]


def test_fastq_fuzz_against_reference_extension(oracle):
    """Live: 94 single characters + 3000 random strings against oracle/_ref/_fastq."""
    import importlib.util
    d = os.path.join(os.path.dirname(HERE), "oracle", "_ref")
    so = [f for f in (os.listdir(d) if os.path.isdir(d) else []) if f.startswith("_fastq.")]
    if not so:
        pytest.skip("oracle/_ref not built")
    spec = importlib.util.spec_from_file_location("_fastq", os.path.join(d, so[0]))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for c in range(33, 127):
        assert oracle.average_error_rate(chr(c)).hex() == ref.average_error_rate(chr(c)).hex()
    rng = random.Random(38)
    for _ in range(3000):
        off = rng.choice([33, 33, 33, 0, 64])
        n = rng.randint(1, 320)
        s = "".join(chr(rng.randint(off, 126)) for _ in range(n))
        assert oracle.average_error_rate(s, phred_offset=off).hex() == \
            ref.average_error_rate(s, phred_offset=off).hex()
    for bad in (" ", "\x7f", "II\x1fI"):
        with pytest.raises(ValueError) as a:
            oracle.average_error_rate(bad)
        with pytest.raises(ValueError) as b:
            ref.average_error_rate(bad)
        assert str(a.value) == str(b.value)
