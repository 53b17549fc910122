"""``fastqdedup._fastq`` (reference _fastqmodule.c, _fastq.pyi)."""
from fastqdedup_amd.cli import average_error_rate  # noqa: F401
