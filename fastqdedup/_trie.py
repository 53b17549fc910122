"""``fastqdedup._trie`` (reference _triemodule.c, _trie.pyi): the device-backed ``Trie``."""
from fastqdedup_amd.core import Trie  # noqa: F401
