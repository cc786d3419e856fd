"""Known-answer tests and constants the reference itself holds for the classify path."""
import pymodel


def test_supermer_kat(orc):
    # T/kmers/minimizer/MinSplitterTest.scala:25-33; MinTable.ofLength(m) is the lexicographic ordering ==
    # identity priority == RandomXOR with mask 0, non-canonical (MinimizerPriorities.scala:246-255).
    p = orc.params(k=5, m=2, spaces=0, xor_mask=0, canonical=False)
    seq = "AATTTACTTTAGTTAC"
    sm = orc.split_encode(p, seq)
    assert [seq[s:s + l] for _, s, l in sm] == ["AATTT", "ATTTA", "TTTACTTT", "CTTTA", "TTTAGTTA", "GTTAC"]


def test_spaced_seed_doc_example(orc):
    # S/kmers/minimizer/MinimizerPriorities.scala:274-277: TTCTGTGGG with s = 3 -> TTCAGAGAG
    p = orc.params(k=9, m=9, spaces=3, xor_mask=0, canonical=False)
    assert orc.decode(orc.priority(p, orc.encode("TTCTGTGGG")), 9) == "TTCAGAGAG"


def test_constants(orc):
    assert orc.DEFAULT_TOGGLE_MASK == 0xe37e28c4271b5a2d  # S/kmers/minimizer/package.scala:32
    assert (orc.AMBIGUOUS_SPAN, orc.MATE_PAIR_BORDER) == (-1, -2)  # S/slacken/package.scala:30-31
    assert (orc.SEQUENCE_FLAG, orc.AMBIGUOUS_FLAG, orc.MATE_PAIR_BORDER_FLAG) == (1, 2, 3)  # :37-39
    assert (orc.NONE, orc.ROOT) == (0, 1)  # S/slacken/Taxonomy.scala:30-31


def test_default_masks(orc):
    # m=31 -> one word, toggle << 2; s=7 leaves 48 significant bits
    p = orc.params()
    assert p.W == 1
    assert p.mask[0] == (orc.DEFAULT_TOGGLE_MASK << 2) & (2**64 - 1)
    assert p.space[0] >> 2 == ((2**62 - 1) & ~0x0CCCCCCC)
    assert bin(p.space[0]).count("1") == 48
    xm, sm = pymodel.masks(31, 7, orc.DEFAULT_TOGGLE_MASK)
    assert (xm << 2, sm << 2) == (p.mask[0] & ~3, p.space[0])


def test_char_classes(orc):
    L = orc.lib()
    for ch, v in zip("ACGTUacgtu", [0, 1, 2, 3, 3] * 2):
        assert L.orc_char_to_twobit(ord(ch)) == v  # BitRepresentation.scala:35-39,127-135
    assert L.orc_char_to_twobit(ord("\n")) == 4 and L.orc_char_to_twobit(ord("\r")) == 4
    for ch in "NnRYKM-. *":
        assert L.orc_char_to_twobit(ord(ch)) == 5


def test_output_strings(orc):
    # TaxonCounts.pairsInOrderString :94-110 / lengthString :114-121 / ClassifiedRead.outputLine Classifier.scala:41-44
    hits = [(5, 3), (5, 2), (0, 4), (-1, 10), (7, 1), (-2, -34), (7, 2), (7, 3)]
    assert orc.pairs_in_order_string(hits) == "5:5 0:4 A:10 7:1 |:| 7:5"
    assert orc.length_string(hits, 35) == "54|39"
    assert orc.length_string(hits[:5], 35) == "54"
    assert orc.output_line(True, "r1", 7, hits[:5], 35) == "C\tr1\t7\t54\t5:5 0:4 A:10 7:1"
    assert orc.pairs_in_order_string([(-2, -34)]) == "|:|" and orc.length_string([(-2, -34)], 35) == "34|34"
