"""``fastqdedup._distance`` (reference _distancemodule.c, _distance.pyi)."""
from fastqdedup_amd.core import within_distance  # noqa: F401
