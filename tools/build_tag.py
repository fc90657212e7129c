"""sha-256 (first 16 hex digits) of the kernel sources a counter pass belongs to: written into profiles/*_traffic.json by the pmc tools and
compared by bench.py, which reports `traffic` only from passes taken on the sources it is running."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRUNK_SOURCES = ("nanocaller_amd/csrc/nc_cnn.hip",)
INDEL_SOURCES = ("nanocaller_amd/csrc/nc_cnn.hip", "nanocaller_amd/csrc/nc_indel.hip", "nanocaller_amd/csrc/nc_pipe.hip")


def build_tag(files, root=ROOT):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
