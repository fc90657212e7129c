"""CPU-only: bench.py reports counter traffic only from passes taken on the kernel sources it runs (tools/build_tag.py)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_traffic_records_of_another_build_are_refused():
    import bench
    from build_tag import INDEL_SOURCES, TRUNK_SOURCES, build_tag
    for name, sources in (("trunk_traffic.json", TRUNK_SOURCES), ("indel_traffic.json", INDEL_SOURCES)):
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert len(rec.get("build_tag", "")) == 16 and rec["build_tag_of"] == list(sources)
        ok, note = bench.traffic_build_check(dict(rec, build_tag=build_tag(sources)), sources)
        assert ok and note is None
        ok, note = bench.traffic_build_check(dict(rec, build_tag="0" * 16), sources)
        assert not ok and "another build" in note
        ok, note = bench.traffic_build_check({k: v for k, v in rec.items() if k != "build_tag"}, sources)      # a record from before the tags
        assert not ok
