"""The banded star alignment as the oracle restates it (oracle.window_band_ref / band_of / nw_cigar_band_free_tail_ref): CPU checks of the
restatement itself -- the GPU tests (tests/test_indel_pipeline.py) compare the device's bands and banded alignments with it."""
import random

from oracle import oracle

SC = (25, 1, 20, -10)


def _mutate(rng, s, p_del=0.03, p_ins=0.03, p_sub=0.05):
    out = []
    for c in s:
        r = rng.random()
        if r < p_del:
            continue
        if r < p_del + p_ins:
            out.append(rng.choice("AGTC"))
        out.append(rng.choice("AGTC") if rng.random() < p_sub else c)
    return "".join(out)


def test_a_band_that_holds_the_optimal_path_gives_the_full_matrix_alignment():
    rng = random.Random(5)
    same = edge = 0
    for _ in range(40):
        ref = "".join(rng.choice("AGTC") for _ in range(161))
        q = _mutate(rng, ref)[:160]
        full = oracle.nw_cigar_free_tail_ref(q, ref, *SC)
        for lo, B in ((-32, 64), (-16, 32), (-8, 32)):
            b = oracle.nw_cigar_band_free_tail_ref(q, ref, lo, B, *SC)
            if b is None:
                edge += 1                                          # the path came to an edge diagonal: the caller aligns on the full matrix
            else:
                assert b == full
                same += 1
    assert same > 80


def test_a_band_too_narrow_for_the_path():
    rng = random.Random(9)
    ref = "".join(rng.choice("AGTC") for _ in range(161))
    q = ref[:40] + ref[60:]                                        # a 20-base deletion: diagonal +20
    full = oracle.nw_cigar_free_tail_ref(q, ref, *SC)
    assert (2, 20) in full
    # hi = 15 < 20: the banded path either runs along the edge diagonal (reported: None) or stays inside with a worse alignment -- which is why
    # the band is derived from the read's own CIGAR (here dmax = 20: 64 diagonals), never guessed
    assert oracle.nw_cigar_band_free_tail_ref(q, ref, -16, 32, *SC) != full
    assert oracle.band_of(0, 20, len(q), len(ref)) == (-22, 64)
    assert oracle.nw_cigar_band_free_tail_ref(q, ref, -22, 64, *SC) == full
    # one lane short of the path: it runs on the edge diagonal
    assert oracle.nw_cigar_band_free_tail_ref(q, ref, -11, 32, *SC) is None


def test_window_band_follows_the_cigar():
    rec = dict(pos0=99, cigar=[("M", 10), ("I", 3), ("M", 5), ("D", 7), ("M", 200), ("S", 2)], seq="A" * 220)
    # anchor at the read's first base: insertion after 10 bases (-3), deletion of 7 after 5 more (+7 -> +4)
    assert oracle.window_band_ref(rec, 100, 160) == (-3, 4)
    # anchor inside the deletion: the window opens on the first base behind it, 4 columns on
    assert oracle.window_band_ref(rec, 118, 160) == (0, 4)
    # a window that ends on the base before the deletion still sees it (the device handles the events of the last base's column)
    assert oracle.window_band_ref(rec, 100, 18) == (-3, 4)
    assert oracle.window_band_ref(rec, 100, 17) == (-3, 0)
    # the soft-clipped tail counts as an insertion behind the last column
    assert oracle.window_band_ref(rec, 300, 160) == (-2, 0)


def test_band_classes():
    assert oracle.band_of(0, 0, 160, 161) == (-16, 32)
    assert oracle.band_of(-3, 4, 160, 161) == (-16, 32)            # w = 7: slack 24, lo = -3 - 12 = -15 -> even
    assert oracle.band_of(0, 30, 160, 161) == (-16, 64)
    assert oracle.band_of(0, 52, 160, 161) is None                 # wider than 64 - 2 * margin
    # a read that ends 50 columns before the window: the band reaches the end of the last row (hi >= n2 - n1)
    lo, B = oracle.band_of(0, 0, 111, 161)
    assert B == 64 and lo + B - 1 >= 50 and lo <= 0
    assert oracle.band_of(0, 0, 60, 161) is None
