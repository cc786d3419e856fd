"""ctypes wrapper around oracle/libslacken_oracle.so (the CPU restatement of the reference's classify path).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (slacken_amd/) must never import it.  Parity pin status: see slacken_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslacken_oracle.so")

MAXW = 4
AMBIGUOUS_SPAN = -1
MATE_PAIR_BORDER = -2
SEQUENCE_FLAG = 1
AMBIGUOUS_FLAG = 2
MATE_PAIR_BORDER_FLAG = 3
NONE = 0
ROOT = 1
DEFAULT_TOGGLE_MASK = 0xE37E28C4271B5A2D


class Params(C.Structure):
    _fields_ = [("k", C.c_int), ("m", C.c_int), ("spaces", C.c_int), ("canonical", C.c_int), ("W", C.c_int),
                ("xor_mask", C.c_uint64), ("mask", C.c_uint64 * MAXW), ("space", C.c_uint64 * MAXW)]


class Span(C.Structure):
    _fields_ = [("key", C.c_uint64 * MAXW), ("kmers", C.c_int32), ("flag", C.c_int32), ("ordinal", C.c_int32),
                ("distinct", C.c_int32)]


class Supermer(C.Structure):
    _fields_ = [("key", C.c_uint64 * MAXW), ("start", C.c_int32), ("length", C.c_int32)]


class Hit(C.Structure):
    _fields_ = [("taxon", C.c_int32), ("count", C.c_int32)]


class ReadResult(C.Structure):
    _fields_ = [("taxon", C.c_int32), ("classified", C.c_int32), ("num_distinct", C.c_int32),
                ("total_kmers", C.c_int32), ("num_hits", C.c_int32)]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "slacken_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64p, i32p, i64p, u8p = (C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                 C.POINTER(C.c_uint8))
        L.orc_params_init.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int]
        L.orc_char_to_twobit.argtypes = [C.c_int]
        L.orc_priority.argtypes = [C.POINTER(Params), u64p, u64p]
        L.orc_priority.restype = None
        L.orc_all_matches.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int, u64p, C.POINTER(C.c_uint8)]
        L.orc_encode.argtypes = [C.c_char_p, C.c_int, u64p]
        L.orc_encode.restype = None
        L.orc_reverse_complement.argtypes = [u64p, C.c_int, u64p]
        L.orc_reverse_complement.restype = None
        L.orc_canonical.argtypes = [u64p, C.c_int, u64p]
        L.orc_canonical.restype = None
        L.orc_split_encode.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int, C.POINTER(Supermer), C.c_int]
        L.orc_split_by_ambiguity.argtypes = [C.c_char_p, C.c_int, C.c_int, i32p, i32p, i32p, C.c_int]
        L.orc_spans.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Span), C.c_int]
        L.orc_minimizer_keys.argtypes = [C.POINTER(Params), C.c_char_p, C.c_long, i64p, C.c_long]
        L.orc_minimizer_keys.restype = C.c_long
        L.orc_library_minimizers.argtypes = [C.POINTER(Params), C.c_char_p, C.c_long, i64p, C.c_long]
        L.orc_library_minimizers.restype = C.c_long
        L.orc_build_records.argtypes = [C.POINTER(Params), i32p, C.c_int32, C.c_char_p, u64p, i32p, C.c_long, i64p, i32p,
                                        C.c_long]
        L.orc_build_records.restype = C.c_long
        L.orc_index_create.argtypes = [C.c_int, i64p, i32p, C.c_size_t]
        L.orc_index_create.restype = C.c_void_p
        L.orc_index_destroy.argtypes = [C.c_void_p]
        L.orc_index_destroy.restype = None
        L.orc_index_lookup.argtypes = [C.c_void_p, u64p]
        L.orc_index_lookup.restype = C.c_int32
        L.orc_lca.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_lca.restype = C.c_int32
        L.orc_resolve_tree.argtypes = [i32p, C.c_int32, i32p, i32p, C.c_int, C.c_double]
        L.orc_resolve_tree.restype = C.c_int32
        L.orc_classify_read.argtypes = [C.POINTER(Params), C.c_void_p, i32p, C.c_int32, C.c_char_p, C.c_int,
                                        C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(ReadResult),
                                        C.POINTER(Hit), C.c_int]
        L.orc_classify_hits.argtypes = [i32p, C.c_int32, C.POINTER(Hit), u8p, C.c_int, C.c_int, C.c_double,
                                        C.POINTER(ReadResult)]
        L.orc_length_string.argtypes = [C.POINTER(Hit), C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_pairs_in_order_string.argtypes = [C.POINTER(Hit), C.c_int, C.c_char_p, C.c_int]
        L.orc_classify_batch.argtypes = [C.POINTER(Params), C.c_void_p, i32p, C.c_int32, u8p, u64p, u8p, u64p,
                                         C.c_size_t, C.c_int, C.POINTER(C.c_double), C.c_int, i32p, u8p, i32p, i32p,
                                         i32p]
        _lib = L
    return _lib


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None else None


def params(k=35, m=31, spaces=7, xor_mask=DEFAULT_TOGGLE_MASK, canonical=True):
    p = Params()
    rc = lib().orc_params_init(C.byref(p), k, m, spaces, C.c_uint64(xor_mask & (2**64 - 1)), int(canonical))
    if rc != 0:
        raise ValueError(f"orc_params_init failed: {rc}")
    return p


def _b(s):
    return s if isinstance(s, bytes) else s.encode("latin-1")


def encode(s):
    s = _b(s)
    out = (C.c_uint64 * MAXW)()
    lib().orc_encode(s, len(s), out)
    return list(out)[:max(1, (len(s) + 31) // 32)]


def decode(words, n):
    return "".join("ACGT"[(words[i // 32] >> (2 * (31 - i % 32))) & 3] for i in range(n))


def reverse_complement(words, size):
    a = (C.c_uint64 * MAXW)(*words)
    out = (C.c_uint64 * MAXW)()
    lib().orc_reverse_complement(a, size, out)
    return list(out)[:len(words)]


def canonical(words, size):
    a = (C.c_uint64 * MAXW)(*words)
    out = (C.c_uint64 * MAXW)()
    lib().orc_canonical(a, size, out)
    return list(out)[:len(words)]


def priority(p, words):
    a = (C.c_uint64 * MAXW)(*words)
    out = (C.c_uint64 * MAXW)()
    lib().orc_priority(C.byref(p), a, out)
    return list(out)[:p.W]


def all_matches(p, seq):
    """ShiftScanner.allMatches (ShiftScanner.scala:90-159): [(key words, valid)] per valid character of seq, or None on an invalid one"""
    b = _b(seq)
    keys = (C.c_uint64 * (max(len(b), 1) * MAXW))()
    valid = (C.c_uint8 * max(len(b), 1))()
    n = lib().orc_all_matches(C.byref(p), b, len(b), keys, valid)
    if n < 0:
        return None
    return [(tuple(keys[i * p.W + j] for j in range(p.W)), bool(valid[i])) for i in range(n)]


def split_encode(p, seq):
    """MinSplitter.splitEncode -> list of (key words tuple, start, length)."""
    seq = _b(seq)
    cap = len(seq) + 2
    out = (Supermer * cap)()
    n = lib().orc_split_encode(C.byref(p), seq, len(seq), out, cap)
    if n < 0:
        raise ValueError(f"orc_split_encode failed: {n}")
    return [(tuple(out[i].key[:p.W]), out[i].start, out[i].length) for i in range(n)]


def split_by_ambiguity(seq, k):
    seq = _b(seq)
    cap = len(seq) + 2
    st, ln, fl = (np.zeros(cap, np.int32) for _ in range(3))
    n = lib().orc_split_by_ambiguity(seq, len(seq), k, _p(st, C.c_int32), _p(ln, C.c_int32), _p(fl, C.c_int32), cap)
    return [(int(st[i]), int(ln[i]), int(fl[i])) for i in range(n)]


def spans(p, seq1, seq2=None):
    """Supermers.spans(splitFragment(..)) -> list of dict(key, kmers, flag, ordinal, distinct)."""
    seq1 = _b(seq1)
    seq2b = _b(seq2) if seq2 is not None else None
    cap = len(seq1) + (len(seq2b) if seq2b is not None else 0) + 4
    out = (Span * cap)()
    n = lib().orc_spans(C.byref(p), seq1, len(seq1), seq2b, len(seq2b) if seq2b is not None else 0, out, cap)
    if n < 0:
        raise ValueError(f"orc_spans failed: {n}")
    return [dict(key=tuple(out[i].key[:p.W]), kmers=out[i].kmers, flag=out[i].flag, ordinal=out[i].ordinal,
                 distinct=bool(out[i].distinct)) for i in range(n)]


def minimizer_keys(p, seq):
    """int64 array of the SEQUENCE-span minimizers (id1) of one sequence, in order."""
    seq = _b(seq)
    out = np.zeros(max(1, len(seq)), np.int64)
    n = lib().orc_minimizer_keys(C.byref(p), seq, len(seq), _p(out, C.c_int64), len(out))
    if n < 0:
        raise ValueError(f"orc_minimizer_keys failed: {n}")
    return out[:n]


def library_minimizers(p, seq):
    """SplitterMinimizers.find for one library sequence (after removeInvalid): int64 minimizers, one per super-mer."""
    seq = _b(seq)
    out = np.zeros(max(1, len(seq)), np.int64)
    n = lib().orc_library_minimizers(C.byref(p), seq, len(seq), _p(out, C.c_int64), len(out))
    if n < 0:
        raise ValueError(f"orc_library_minimizers failed: {n}")
    return out[:n]


def build_records(p, parents, bases, offsets, taxa):
    """KeyValueIndex.makeRecords: (keys sorted ascending, LCA taxa) of taxon-labelled sequences (same argument meaning as
    slk_index_add_sequences)."""
    parents = np.ascontiguousarray(parents, np.int32)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    taxa = np.ascontiguousarray(taxa, np.int32)
    cap = int(offsets[-1]) + 1
    keys = np.zeros(cap, np.int64)
    tx = np.zeros(cap, np.int32)
    n = lib().orc_build_records(C.byref(p), _p(parents, C.c_int32), len(parents), bases.tobytes(), _p(offsets, C.c_uint64),
                                _p(taxa, C.c_int32), len(taxa), _p(keys, C.c_int64), _p(tx, C.c_int32), cap)
    if n < 0:
        raise ValueError(f"orc_build_records failed: {n}")
    return keys[:n].copy(), tx[:n].copy()


class Index:
    """The records side of the reference's join: (id1..idW: int64, taxon: int32)."""

    def __init__(self, W, keys, taxa):
        self.W = W
        self.keys = np.ascontiguousarray(keys, dtype=np.int64).reshape(-1, W)
        self.taxa = np.ascontiguousarray(taxa, dtype=np.int32)
        assert len(self.keys) == len(self.taxa)
        self.h = lib().orc_index_create(W, _p(self.keys, C.c_int64), _p(self.taxa, C.c_int32), len(self.taxa))

    def lookup(self, key_words):
        a = (C.c_uint64 * MAXW)(*[w & (2**64 - 1) for w in key_words])
        return lib().orc_index_lookup(self.h, a)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_index_destroy(self.h)
            self.h = None


def lca(parents, a, b):
    parents = np.ascontiguousarray(parents, np.int32)
    return lib().orc_lca(_p(parents, C.c_int32), len(parents), a, b)


def resolve_tree(parents, map_taxa, map_counts, required_score):
    parents = np.ascontiguousarray(parents, np.int32)
    t = np.ascontiguousarray(map_taxa, np.int32)
    c = np.ascontiguousarray(map_counts, np.int32)
    return lib().orc_resolve_tree(_p(parents, C.c_int32), len(parents), _p(t, C.c_int32), _p(c, C.c_int32), len(t),
                                  float(required_score))


def classify_read(p, index, parents, seq1, seq2=None, min_hit_groups=2, confidence=0.0):
    """Returns (result dict, hits [(taxon, count)...] un-merged in ordinal order)."""
    parents = np.ascontiguousarray(parents, np.int32)
    seq1 = _b(seq1)
    seq2b = _b(seq2) if seq2 is not None else None
    cap = len(seq1) + (len(seq2b) if seq2b is not None else 0) + 4
    hits = (Hit * cap)()
    res = ReadResult()
    n = lib().orc_classify_read(C.byref(p), index.h, _p(parents, C.c_int32), len(parents), seq1, len(seq1), seq2b,
                                len(seq2b) if seq2b is not None else 0, min_hit_groups, float(confidence),
                                C.byref(res), hits, cap)
    if n < 0:
        raise ValueError(f"orc_classify_read failed: {n}")
    r = dict(taxon=res.taxon, classified=bool(res.classified), num_distinct=res.num_distinct,
             total_kmers=res.total_kmers, num_hits=res.num_hits)
    return r, [(hits[i].taxon, hits[i].count) for i in range(n)]


def classify_hits(parents, hits, distinct, min_hit_groups=2, confidence=0.0):
    """Classifier.classify (Classifier.scala:439-454) on a given hit list [(taxon, count)...] in ordinal order with the
    spans' distinct flags -- e.g. the list merged from the fragments that share a title (Classifier.scala:92,136)."""
    parents = np.ascontiguousarray(parents, np.int32)
    arr = (Hit * max(1, len(hits)))(*[Hit(t, c) for t, c in hits])
    d = np.ascontiguousarray(list(distinct) + [0], np.uint8)
    res = ReadResult()
    rc = lib().orc_classify_hits(_p(parents, C.c_int32), len(parents), arr, _p(d, C.c_uint8), len(hits), min_hit_groups,
                                 float(confidence), C.byref(res))
    assert rc == 0
    return dict(taxon=res.taxon, classified=bool(res.classified), num_distinct=res.num_distinct,
                total_kmers=res.total_kmers, num_hits=res.num_hits)


def length_string(hits, k):
    arr = (Hit * max(1, len(hits)))(*[Hit(t, c) for t, c in hits])
    buf = C.create_string_buffer(64)
    n = lib().orc_length_string(arr, len(hits), k, buf, 64)
    assert n >= 0
    return buf.value.decode()


def pairs_in_order_string(hits):
    arr = (Hit * max(1, len(hits)))(*[Hit(t, c) for t, c in hits])
    cap = 32 * len(hits) + 16
    buf = C.create_string_buffer(cap)
    n = lib().orc_pairs_in_order_string(arr, len(hits), buf, cap)
    assert n >= 0
    return buf.value.decode()


def output_line(classified, title, taxon, hits, k):
    """ClassifiedRead.outputLine, Classifier.scala:41-44."""
    return f"{'C' if classified else 'U'}\t{title}\t{taxon}\t{length_string(hits, k)}\t{pairs_in_order_string(hits)}"


def classify_batch(p, index, parents, bases, offsets, mate_bases=None, mate_offsets=None, min_hit_groups=2,
                   thresholds=(0.0,)):
    """Same argument meaning as slk_classify_batch. Returns dict of numpy arrays (+ 'threads')."""
    parents = np.ascontiguousarray(parents, np.int32)
    bases = np.ascontiguousarray(bases, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.uint64)
    R = len(offsets) - 1
    Cn = len(thresholds)
    thr = (C.c_double * Cn)(*thresholds)
    out_taxon = np.zeros((Cn, R), np.int32)
    out_cls = np.zeros((Cn, R), np.uint8)
    nd, tk, nh = (np.zeros(R, np.int32) for _ in range(3))
    if mate_bases is not None:
        mate_bases = np.ascontiguousarray(mate_bases, np.uint8)
        mate_offsets = np.ascontiguousarray(mate_offsets, np.uint64)
    rc = lib().orc_classify_batch(C.byref(p), index.h, _p(parents, C.c_int32), len(parents), _p(bases, C.c_uint8),
                                  _p(offsets, C.c_uint64), _p(mate_bases, C.c_uint8), _p(mate_offsets, C.c_uint64),
                                  R, min_hit_groups, thr, Cn, _p(out_taxon, C.c_int32), _p(out_cls, C.c_uint8),
                                  _p(nd, C.c_int32), _p(tk, C.c_int32), _p(nh, C.c_int32))
    if rc < 0:
        raise ValueError(f"orc_classify_batch failed: {rc}")
    return dict(taxon=out_taxon, classified=out_cls, num_distinct=nd, total_kmers=tk, num_hits=nh, threads=rc)
