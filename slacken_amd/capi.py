"""ctypes binding of include/slacken_amd.h (one Python method per C entry point; no compute happens in Python).
In a process that also uses PyTorch-ROCm, `import torch` before the first slacken_amd call: torch bundles its own HIP runtime,
and whichever runtime is loaded first is the one that can open the GPU."""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULT_TOGGLE_MASK = 0xE37E28C4271B5A2D
E_INVALID, E_UNSUPPORTED, E_HIP, E_NO_GPU, E_CAPACITY, E_STATE = -1, -2, -3, -4, -5, -6   # SLK_E_* of the header
TAXON_NONE, TAXON_ROOT, TAXON_AMBIGUOUS, TAXON_MATE_PAIR_BORDER = 0, 1, -1, -2
FLAG_SEQUENCE, FLAG_AMBIGUOUS, FLAG_MATE_PAIR_BORDER = 1, 2, 3


class SlackenError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"slacken_amd error {code}: {msg}")
        self.code = code


class _Params(C.Structure):
    _fields_ = [("k", C.c_int32), ("m", C.c_int32), ("spaces", C.c_int32), ("canonical", C.c_int32),
                ("xor_mask", C.c_uint64), ("id_longs", C.c_int32), ("reserved", C.c_int32)]


class _TableConfig(C.Structure):
    _fields_ = [("expected_records", C.c_uint64), ("max_taxon", C.c_int32), ("load_factor", C.c_float)]


class IndexInfo(C.Structure):
    _fields_ = [("records", C.c_uint64), ("buckets", C.c_uint64), ("table_bytes", C.c_uint64),
                ("bucket_bits", C.c_int32), ("taxon_bits", C.c_int32), ("disp_bits", C.c_int32),
                ("max_displacement", C.c_int32), ("duplicate_keys", C.c_uint64), ("taxonomy_size", C.c_int32),
                ("device", C.c_int32), ("dense_taxa", C.c_int32), ("bucket_cells", C.c_int32), ("load_factor", C.c_float),
                ("grown", C.c_int32)]


class _ShardBatch(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("mate_bases", C.c_void_p), ("mate_offsets", C.c_void_p),
                ("R", C.c_uint64), ("out_taxon", C.c_void_p), ("out_classified", C.c_void_p), ("out_num_distinct", C.c_void_p),
                ("out_total_kmers", C.c_void_p), ("out_hit_offsets", C.c_void_p), ("out_hits", C.c_void_p),
                ("hits_capacity", C.c_uint64)]


class ShardLists(C.Structure):
    """slk_shard_lists: a batch and the lists its EMIT job leaves on its rank (all device addresses)"""
    _fields_ = [("d_bases", C.c_void_p), ("d_offsets", C.c_void_p), ("d_mate_bases", C.c_void_p), ("d_mate_offsets", C.c_void_p),
                ("R", C.c_uint64), ("total_bases", C.c_uint64), ("total_mate_bases", C.c_uint64), ("n_shards", C.c_uint32),
                ("reserved", C.c_uint32), ("capacity_per_owner", C.c_uint64),
                ("d_send_keys", C.c_void_p), ("d_send_meta", C.c_void_p), ("d_cursors", C.c_void_p), ("d_batch_log", C.c_void_p),
                ("d_tile_rows", C.c_void_p), ("d_read_info", C.c_void_p), ("d_defer", C.c_void_p), ("d_span_meta", C.c_void_p),
                ("d_span_taxon", C.c_void_p), ("d_span_count", C.c_void_p)]


class ShardLookup(C.Structure):
    _fields_ = [("d_keys", C.c_void_p), ("n", C.c_uint64), ("d_out_taxa", C.c_void_p)]


class ShardResults(C.Structure):
    _fields_ = [("d_taxa", C.c_void_p), ("min_hit_groups", C.c_int32), ("C", C.c_int32), ("thresholds", C.POINTER(C.c_double)),
                ("d_out_taxon", C.c_void_p), ("d_out_classified", C.c_void_p), ("d_out_num_distinct", C.c_void_p),
                ("d_out_total_kmers", C.c_void_p), ("d_out_num_hits", C.c_void_p)]


EXCHANGE_AUTO, EXCHANGE_RCCL, EXCHANGE_COPY = 0, 1, 2
SPAN_DTYPE = np.dtype([("key", "<i8"), ("kmers", "<i4"), ("flag", "i1"), ("distinct", "u1"), ("pad", "<u2")])
HIT_DTYPE = np.dtype([("taxon", "<i4"), ("count", "<i4")])

# every symbol include/slacken_amd.h declares
EXPORTS = ["slk_device_count", "slk_last_error", "slk_version", "slk_host_alloc", "slk_host_register", "slk_host_free", "slk_index_create", "slk_index_append",
           "slk_index_append_device", "slk_index_set_shard", "slk_index_set_taxonomy", "slk_index_finalize", "slk_index_get_info",
           "slk_index_lookup", "slk_index_add_sequences", "slk_index_add_sequences_device", "slk_index_export", "slk_index_destroy", "slk_stream_create", "slk_stream_synchronize",
           "slk_stream_hip_stream", "slk_stream_destroy", "slk_spans_batch", "slk_spans_batch_wide", "slk_classify_batch",
           "slk_classify_batch_packed", "slk_pack_bases",
           "slk_classify_batch_device", "slk_classify_hits", "slk_stream_last_stage_ms", "slk_scan_device", "slk_lookup_device",
           "slk_shard_of", "slk_stream_set_merged_hits", "slk_classify_hits_device", "slk_shard_batch_rows", "slk_shard_chunk", "slk_shard_step_device",
           "slk_stream_last_deferred", "slk_table_slot", "slk_table_hash_of",
           "slk_shardset_create", "slk_shardset_classify", "slk_shardset_classify_rounds", "slk_shardset_exchange_mode", "slk_shardset_destroy"]


def lib_path():
    # SLACKEN_AMD_LIB: another build of the same library (A/B experiments with different compile-time settings)
    return os.environ.get("SLACKEN_AMD_LIB") or os.path.join(_HERE, "lib", "libslacken_amd.so")


_lib = None


def lib():
    """Load libslacken_amd.so; raises loudly if it has not been built (there is no fallback implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} is missing: build it with `make` (or __graft_entry__.build()). "
                          "slacken_amd has no CPU/Python fallback for the classify path.")
    L = C.CDLL(path)
    vp, u8p, i32p, i64p, u64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.slk_device_count.restype = C.c_int32
    L.slk_last_error.restype = C.c_char_p
    L.slk_version.restype = C.c_char_p
    L.slk_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.slk_host_register.argtypes = [vp, C.c_size_t]
    L.slk_host_free.argtypes = [vp]
    L.slk_index_create.argtypes = [C.POINTER(_Params), C.POINTER(_TableConfig), C.c_int32, C.POINTER(vp)]
    L.slk_index_append.argtypes = [vp, i64p, i32p, C.c_uint64]
    L.slk_index_append_device.argtypes = [vp, i64p, i32p, C.c_uint64]
    L.slk_index_set_taxonomy.argtypes = [vp, i32p, C.c_int32]
    L.slk_table_slot.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    L.slk_table_hash_of.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    L.slk_shardset_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.slk_shardset_classify.argtypes = [vp, C.POINTER(_ShardBatch), C.c_int32, C.POINTER(C.c_double), C.c_int32]
    L.slk_shardset_classify_rounds.argtypes = [vp, C.POINTER(_ShardBatch), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_int32]
    L.slk_shardset_exchange_mode.argtypes = [vp]
    L.slk_shardset_destroy.argtypes = [vp]
    L.slk_shardset_destroy.restype = None
    L.slk_index_set_shard.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.slk_index_finalize.argtypes = [vp]
    L.slk_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    L.slk_index_lookup.argtypes = [vp, i64p, C.c_uint64, i32p]
    L.slk_index_add_sequences.argtypes = [vp, u8p, u64p, i32p, C.c_uint64]
    L.slk_index_add_sequences_device.argtypes = [vp, u8p, u64p, i32p, C.c_uint64]
    L.slk_index_export.argtypes = [vp, i64p, i32p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.slk_index_destroy.argtypes = [vp]
    L.slk_index_destroy.restype = None
    L.slk_stream_create.argtypes = [vp, C.POINTER(vp)]
    L.slk_stream_synchronize.argtypes = [vp]
    L.slk_stream_set_merged_hits.argtypes = [vp, C.c_int32]
    L.slk_stream_hip_stream.argtypes = [vp]
    L.slk_stream_hip_stream.restype = C.c_void_p
    L.slk_stream_destroy.argtypes = [vp]
    L.slk_stream_destroy.restype = None
    L.slk_spans_batch.argtypes = [vp, vp, u8p, u64p, u8p, u64p, C.c_uint64, u64p, vp, C.c_uint64]
    L.slk_spans_batch_wide.argtypes = [vp, vp, u8p, u64p, u8p, u64p, C.c_uint64, u64p, vp, i64p, C.c_uint64]
    L.slk_classify_batch.argtypes = [vp, vp, u8p, u64p, u8p, u64p, C.c_uint64, C.c_int32, C.POINTER(C.c_double),
                                     C.c_int32, i32p, u8p, i32p, i32p, u64p, vp, C.c_uint64]
    L.slk_classify_batch_packed.argtypes = [vp, vp, vp, vp, u64p, vp, vp, u64p, C.c_uint64, C.c_int32, C.POINTER(C.c_double),
                                            C.c_int32, i32p, u8p, i32p, i32p, u64p, vp, C.c_uint64]
    L.slk_pack_bases.argtypes = [u8p, C.c_uint64, vp, vp]
    L.slk_classify_batch_device.argtypes = [vp, vp, u8p, u64p, u8p, u64p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.c_int32, C.POINTER(C.c_double), C.c_int32, i32p, u8p, i32p, i32p, i32p, i32p]
    L.slk_classify_hits.argtypes = [vp, vp, C.c_uint64, u64p, vp, u8p, C.c_int32, C.POINTER(C.c_double), C.c_int32, i32p, u8p,
                                    i32p, i32p]
    L.slk_stream_last_stage_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.slk_scan_device.argtypes = [vp, vp, u8p, u64p, u8p, u64p, C.c_uint64, u64p, i32p, i32p]
    L.slk_lookup_device.argtypes = [vp, vp, i64p, C.c_uint64, i32p]
    L.slk_shard_of.argtypes = [C.c_int64, C.c_uint32]
    L.slk_shard_of.restype = C.c_uint32
    L.slk_classify_hits_device.argtypes = [vp, vp, u64p, u64p, C.c_uint64, i32p, i32p, i32p, u64p, C.c_int32,
                                           C.POINTER(C.c_double), C.c_int32, i32p, u8p, i32p, i32p, i32p]
    L.slk_shard_batch_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int32]
    L.slk_shard_batch_rows.restype = C.c_uint64
    L.slk_shard_chunk.argtypes = [C.c_uint32]
    L.slk_shard_chunk.restype = C.c_uint32
    L.slk_shard_step_device.argtypes = [vp, vp, C.POINTER(ShardLists), C.POINTER(ShardLookup), C.POINTER(ShardLists), C.POINTER(ShardResults)]
    L.slk_stream_last_deferred.argtypes = [vp, C.POINTER(C.c_uint64)]
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int:  # default: int32 status
            fn.restype = C.c_int32
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise SlackenError(rc, lib().slk_last_error().decode())


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    return a.ctypes.data if a is not None else None


def pack_bases(bases, pinned=False):
    """ASCII bases -> (codes uint32 [ceil(n / 16)], valid uint16 [ceil(n / 16)]): the engine's 3-bit form (slk_pack_bases)"""
    bases = _np(bases, np.uint8)
    words = (bases.size + 15) // 16
    make = (lambda n, dt: pinned_array((n,), dt)) if pinned else (lambda n, dt: np.zeros(n, dt))
    codes, valid = make(max(words, 1), np.uint32), make(max(words, 1), np.uint16)
    _check(lib().slk_pack_bases(_ptr(bases), bases.size, _ptr(codes), _ptr(valid)))
    return codes, valid


def pinned_array(shape, dtype):
    """A numpy array in pinned host memory (slk_host_alloc): the host entry points DMA from and to it directly.
    Keep the array (or a view of it) alive while it is in use; the memory is freed with the array's base object."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    p = C.c_void_p()
    _check(lib().slk_host_alloc(max(n, 1), C.byref(p)))

    class _Owner:
        def __init__(self, addr):
            self.addr = addr

        def __del__(self):
            try:
                lib().slk_host_free(self.addr)
            except Exception:
                pass

    buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
    buf._owner = _Owner(p.value)   # freed when the last view goes
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


@dataclass
class ClassifyParams:
    """Mirror of ClassifyParams (S/slacken/Classifier.scala:60-61)."""
    minHitGroups: int = 2
    withUnclassified: bool = True
    thresholds: List[float] = field(default_factory=lambda: [0.0])
    sampleRegex: Optional[str] = None
    perReadOutput: bool = True


class Index:
    """HBM-resident minimizer->taxon records + taxonomy (the engine-side KeyValueIndex)."""

    def __init__(self, k=35, m=31, spaces=7, xor_mask=DEFAULT_TOGGLE_MASK, canonical=True, expected_records=1 << 20,
                 max_taxon=0, load_factor=0.0, device=0):
        self.k, self.m, self.spaces = k, m, spaces
        self.W = (m + 31) // 32   # id columns: keys are rows of W int64 words
        p = _Params(k, m, spaces, int(bool(canonical)), C.c_uint64(xor_mask & (2**64 - 1)), (m + 31) // 32, 0)
        cfg = _TableConfig(int(expected_records), int(max_taxon), float(load_factor))
        h = C.c_void_p()
        _check(lib().slk_index_create(C.byref(p), C.byref(cfg), device, C.byref(h)))
        self.h = h

    def append(self, keys, taxa):
        keys, taxa = _np(keys, np.int64), _np(taxa, np.int32)
        assert keys.size == taxa.size * self.W
        _check(lib().slk_index_append(self.h, _ptr(keys), _ptr(taxa), taxa.size))

    def append_device(self, d_keys_ptr, d_taxa_ptr, n):
        _check(lib().slk_index_append_device(self.h, d_keys_ptr, d_taxa_ptr, n))

    def set_shard(self, shard, n_shards):
        """Table-sharded library: keep only the records whose key falls to `shard` of `n_shards` (before the first record)."""
        _check(lib().slk_index_set_shard(self.h, shard, n_shards))

    def set_taxonomy(self, parents):
        parents = _np(parents, np.int32)
        _check(lib().slk_index_set_taxonomy(self.h, _ptr(parents), parents.size))

    def add_sequences(self, bases, offsets, taxa):
        """Library construction (KeyValueIndex.makeRecords): minimizers of taxon-labelled sequences, LCA-merged."""
        bases, offsets, taxa = _np(bases, np.uint8), _np(offsets, np.uint64), _np(taxa, np.int32)
        assert offsets.size == taxa.size + 1
        _check(lib().slk_index_add_sequences(self.h, _ptr(bases), _ptr(offsets), _ptr(taxa), taxa.size))

    def add_sequences_device(self, d_bases_ptr, offsets, taxa):
        """add_sequences with the bases resident on the index's GPU (raw device address, 16 readable bytes past the end)."""
        offsets, taxa = _np(offsets, np.uint64), _np(taxa, np.int32)
        assert offsets.size == taxa.size + 1
        _check(lib().slk_index_add_sequences_device(self.h, d_bases_ptr, _ptr(offsets), _ptr(taxa), taxa.size))

    def export(self):
        """(keys, taxa) of every record in the table, sorted by key."""
        n = C.c_uint64(0)
        _check(lib().slk_index_export(self.h, None, None, 0, C.byref(n)))
        keys, taxa = np.zeros(n.value * self.W, np.int64), np.zeros(n.value, np.int32)
        if n.value:
            _check(lib().slk_index_export(self.h, _ptr(keys), _ptr(taxa), n.value, C.byref(n)))
        if self.W > 1:   # rows of W words, sorted as unsigned numbers, word by word
            rows = keys.reshape(-1, self.W)
            order = np.lexsort(rows.view(np.uint64).T[::-1])
            return rows[order], taxa[order]
        order = np.argsort(keys, kind="stable")
        return keys[order], taxa[order]

    def finalize(self):
        _check(lib().slk_index_finalize(self.h))

    def info(self):
        out = IndexInfo()
        _check(lib().slk_index_get_info(self.h, C.byref(out)))
        return out

    def lookup(self, keys):
        keys = _np(keys, np.int64)
        out = np.zeros(keys.size // self.W, np.int32)
        _check(lib().slk_index_lookup(self.h, _ptr(keys), out.size, _ptr(out)))
        return out

    def stream(self):
        return Stream(self)

    def close(self):
        if getattr(self, "h", None):
            lib().slk_index_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardSet:
    """Table-sharded classification in one process (slk_shardset_*): members[g] is an Index with set_shard(g, n), finalized."""

    def __init__(self, members, exchange=EXCHANGE_AUTO):
        self.members = list(members)
        arr = (C.c_void_p * len(self.members))(*[m.h for m in self.members])
        h = C.c_void_p()
        _check(lib().slk_shardset_create(arr, len(self.members), exchange, C.byref(h)))
        self.h = h

    @property
    def exchange_mode(self):
        return int(lib().slk_shardset_exchange_mode(self.h))

    def classify(self, batches, min_hit_groups=2, thresholds=(0.0,), with_hits=True):
        """One round: batches[g] = (bases, offsets) or (bases, offsets, mate_bases, mate_offsets) or None for member g.
        -> list of result dicts as Stream.classify_batch returns them (None for members without a batch)."""
        n = len(self.members)
        assert len(batches) == n
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        arr = (_ShardBatch * n)()
        keep, outs = [], []
        for g, b in enumerate(batches):
            if b is None:
                outs.append(None)
                continue
            bases, offsets = _np(b[0], np.uint8), _np(b[1], np.uint64)
            mb = mo = None
            if len(b) > 2 and b[2] is not None:
                mb, mo = _np(b[2], np.uint8), _np(b[3], np.uint64)
            R = offsets.size - 1
            o = dict(taxon=np.zeros((Cn, R), np.int32), classified=np.zeros((Cn, R), np.uint8), num_distinct=np.zeros(R, np.int32),
                     total_kmers=np.zeros(R, np.int32))
            hit_off = hits = None
            cap = 0
            if with_hits:
                cap = int(bases.size + (mb.size + R if mb is not None else 0)) + 1
                hit_off, hits = np.zeros(R + 1, np.uint64), np.zeros(cap, HIT_DTYPE)
            keep.append((bases, offsets, mb, mo, hit_off, hits))
            arr[g] = _ShardBatch(_ptr(bases), _ptr(offsets), _ptr(mb), _ptr(mo), R, _ptr(o["taxon"]), _ptr(o["classified"]),
                                 _ptr(o["num_distinct"]), _ptr(o["total_kmers"]), _ptr(hit_off), _ptr(hits), cap)
            o["_hit"] = (hit_off, hits)
            outs.append(o)
        _check(lib().slk_shardset_classify(self.h, arr, min_hit_groups, thr, Cn))
        for o in outs:
            if o is None:
                continue
            hit_off, hits = o.pop("_hit")
            if hit_off is not None:
                o["hit_offsets"] = hit_off
                o["hits"] = hits[:int(hit_off[-1])]
                o["num_hits"] = np.diff(hit_off.astype(np.int64)).astype(np.int32)
        return outs

    def classify_rounds_device(self, rounds, min_hit_groups=2, thresholds=(0.0,)):
        """Pipelined rounds over DEVICE-resident batches (slk_shardset_classify_rounds, device_resident = 1): rounds[r][g] is a dict of
        raw device addresses on member g's GPU -- bases, offsets, R, out_taxon, out_classified and optionally mate_bases,
        mate_offsets, out_num_distinct, out_total_kmers -- or None.  Synchronous: returns when every round is done."""
        n = len(self.members)
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        arr = (_ShardBatch * (n * len(rounds)))()
        for r, rnd in enumerate(rounds):
            assert len(rnd) == n
            for g, b in enumerate(rnd):
                if b is None:
                    continue
                arr[r * n + g] = _ShardBatch(b["bases"], b["offsets"], b.get("mate_bases"), b.get("mate_offsets"), b["R"], b["out_taxon"],
                                             b["out_classified"], b.get("out_num_distinct"), b.get("out_total_kmers"), None, None, 0)
        _check(lib().slk_shardset_classify_rounds(self.h, arr, len(rounds), 1, min_hit_groups, thr, Cn))

    def classify_rounds(self, rounds, min_hit_groups=2, thresholds=(0.0,), with_hits=True):
        """Several rounds of classify() in ONE pipelined call (slk_shardset_classify_rounds, host pointers): rounds[r][g] as
        classify()'s batches[g].  -> list (per round) of lists (per member) of result dicts."""
        n = len(self.members)
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        arr = (_ShardBatch * (n * len(rounds)))()
        keep, outs = [], []
        for r, rnd in enumerate(rounds):
            assert len(rnd) == n
            row = []
            for g, b in enumerate(rnd):
                if b is None:
                    row.append(None)
                    continue
                bases, offsets = _np(b[0], np.uint8), _np(b[1], np.uint64)
                mb = mo = None
                if len(b) > 2 and b[2] is not None:
                    mb, mo = _np(b[2], np.uint8), _np(b[3], np.uint64)
                R = offsets.size - 1
                o = dict(taxon=np.zeros((Cn, R), np.int32), classified=np.zeros((Cn, R), np.uint8), num_distinct=np.zeros(R, np.int32),
                         total_kmers=np.zeros(R, np.int32))
                hit_off = hits = None
                cap = 0
                if with_hits:
                    cap = int(bases.size + (mb.size + R if mb is not None else 0)) + 1
                    hit_off, hits = np.zeros(R + 1, np.uint64), np.zeros(cap, HIT_DTYPE)
                keep.append((bases, offsets, mb, mo, hit_off, hits))
                arr[r * n + g] = _ShardBatch(_ptr(bases), _ptr(offsets), _ptr(mb), _ptr(mo), R, _ptr(o["taxon"]), _ptr(o["classified"]),
                                             _ptr(o["num_distinct"]), _ptr(o["total_kmers"]), _ptr(hit_off), _ptr(hits), cap)
                o["_hit"] = (hit_off, hits)
                row.append(o)
            outs.append(row)
        _check(lib().slk_shardset_classify_rounds(self.h, arr, len(rounds), 0, min_hit_groups, thr, Cn))
        for row in outs:
            for o in row:
                if o is None:
                    continue
                hit_off, hits = o.pop("_hit")
                if hit_off is not None:
                    o["hit_offsets"] = hit_off
                    o["hits"] = hits[:int(hit_off[-1])]
                    o["num_hits"] = np.diff(hit_off.astype(np.int64)).astype(np.int32)
        return outs

    def close(self):
        if getattr(self, "h", None):
            lib().slk_shardset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_deferred_streams = []   # engine streams whose destruction was put off because torch still held a view of them (Stream.close)


class Stream:
    def __init__(self, index):
        self.index = index
        h = C.c_void_p()
        _check(lib().slk_stream_create(index.h, C.byref(h)))
        self.h = h
        self._views = []     # weak references to the torch ExternalStream wrappers handed out

    def external_stream(self, torch, device):
        """The torch view of this stream (torch.cuda.ExternalStream).  torch's caching allocator ties every block to the stream it was
        allocated on and records events on the streams a tensor was used on when the tensor dies -- on a stream that no longer
        exists that is a crash -- so the engine stream is NOT destroyed while a view of it is alive: close() puts the destruction
        off (see below), and whoever hands views out drops them, after torch.cuda.empty_cache(), before it closes the stream
        (slacken_amd.sharded.ShardedClassifier.close)."""
        import weakref
        ext = torch.cuda.ExternalStream(self.hip_stream, device=device)
        self._views.append(weakref.ref(ext))
        return ext

    def views_alive(self):
        return any(r() is not None for r in self._views)

    @property
    def hip_stream(self):
        return lib().slk_stream_hip_stream(self.h)

    def synchronize(self):
        _check(lib().slk_stream_synchronize(self.h))

    def set_merged_hits(self, on=True):
        """hit lists of classify_batch as TaxonCounts.fromHits merges them (adjacent hits of one taxon summed); `num_hits` then counts
        the merged entries"""
        _check(lib().slk_stream_set_merged_hits(self.h, 1 if on else 0))

    def last_deferred(self):
        n = C.c_uint64(0)
        _check(lib().slk_stream_last_deferred(self.h, C.byref(n)))
        return int(n.value)

    def spans_batch(self, bases, offsets, mate_bases=None, mate_offsets=None, capacity=None):
        """-> (span_offsets u64[R+1], spans structured array SPAN_DTYPE)"""
        bases, offsets = _np(bases, np.uint8), _np(offsets, np.uint64)
        R = offsets.size - 1
        if mate_bases is not None:
            mate_bases, mate_offsets = _np(mate_bases, np.uint8), _np(mate_offsets, np.uint64)
        if capacity is None:
            capacity = int(bases.size + (mate_bases.size + R if mate_bases is not None else 0)) + 1
        out_off = np.zeros(R + 1, np.uint64)
        out = np.zeros(capacity, SPAN_DTYPE)
        _check(lib().slk_spans_batch(self.index.h, self.h, _ptr(bases), _ptr(offsets), _ptr(mate_bases),
                                     _ptr(mate_offsets), R, _ptr(out_off), _ptr(out), capacity))
        return out_off, out[:int(out_off[R])]

    def spans_batch_wide(self, bases, offsets, mate_bases=None, mate_offsets=None):
        """-> (span_offsets u64[R+1], spans SPAN_DTYPE, keys int64 [n_spans, id_longs])"""
        bases, offsets = _np(bases, np.uint8), _np(offsets, np.uint64)
        R = offsets.size - 1
        if mate_bases is not None:
            mate_bases, mate_offsets = _np(mate_bases, np.uint8), _np(mate_offsets, np.uint64)
        capacity = int(bases.size + (mate_bases.size + R if mate_bases is not None else 0)) + 1
        out_off = np.zeros(R + 1, np.uint64)
        out = np.zeros(capacity, SPAN_DTYPE)
        keys = np.zeros((capacity, self.index.W), np.int64)
        _check(lib().slk_spans_batch_wide(self.index.h, self.h, _ptr(bases), _ptr(offsets), _ptr(mate_bases), _ptr(mate_offsets), R,
                                          _ptr(out_off), _ptr(out), _ptr(keys), capacity))
        n = int(out_off[R])
        return out_off, out[:n], keys[:n]

    def classify_batch(self, bases, offsets, mate_bases=None, mate_offsets=None, min_hit_groups=2,
                       thresholds=(0.0,), with_hits=True, hits_capacity=None, with_num_hits=False, out=None, packed=False):
        """out: optional dict of preallocated result arrays (taxon [C,R] int32, classified [C,R] uint8, num_distinct [R],
        total_kmers [R]) -- e.g. pinned_array()s, which the library fills by DMA.
        packed: True = the reads travel in the engine's 3-bit form (packed here by slk_pack_bases; slk_classify_batch_packed), or a
        tuple (codes, valid[, mate_codes, mate_valid]) of arrays packed beforehand (bases may then be None)."""
        offsets = _np(offsets, np.uint64)
        R = offsets.size - 1
        pk = None
        if packed:
            if packed is True:
                pk = pack_bases(bases) + (pack_bases(mate_bases) if mate_bases is not None else (None, None))
            else:
                pk = tuple(packed) + (None, None) * (len(packed) == 2)
            n_bases = int(offsets[R])
            n_mates = int(_np(mate_offsets, np.uint64)[R]) if mate_offsets is not None else 0
        else:
            bases = _np(bases, np.uint8)
            n_bases, n_mates = bases.size, (np.asarray(mate_bases).size if mate_bases is not None else 0)
        if mate_offsets is not None:
            mate_offsets = _np(mate_offsets, np.uint64)
            if not packed:
                mate_bases = _np(mate_bases, np.uint8)
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        if out is not None:
            taxon, cls, nd, tk = out["taxon"], out["classified"], out["num_distinct"], out["total_kmers"]
            assert taxon.shape == (Cn, R) and taxon.dtype == np.int32 and cls.shape == (Cn, R) and cls.dtype == np.uint8
            assert all(a.flags.c_contiguous for a in (taxon, cls, nd, tk))
        else:
            taxon = np.zeros((Cn, R), np.int32)
            cls = np.zeros((Cn, R), np.uint8)
            nd, tk = np.zeros(R, np.int32), np.zeros(R, np.int32)
        hit_off = hits = None
        cap = 0
        if with_hits:
            cap = hits_capacity if hits_capacity is not None else \
                int(n_bases + (n_mates + R if mate_offsets is not None else 0)) + 1
            hit_off = np.zeros(R + 1, np.uint64)
            hits = np.zeros(cap, HIT_DTYPE)
        elif with_num_hits:   # the number of spans per fragment without the lists: offsets only
            hit_off = np.zeros(R + 1, np.uint64)
        if pk is not None:
            _check(lib().slk_classify_batch_packed(self.index.h, self.h, _ptr(pk[0]), _ptr(pk[1]), _ptr(offsets), _ptr(pk[2]), _ptr(pk[3]),
                                                   _ptr(mate_offsets), R, min_hit_groups, thr, Cn, _ptr(taxon), _ptr(cls),
                                                   _ptr(nd), _ptr(tk), _ptr(hit_off), _ptr(hits), cap))
        else:
            _check(lib().slk_classify_batch(self.index.h, self.h, _ptr(bases), _ptr(offsets), _ptr(mate_bases),
                                            _ptr(mate_offsets), R, min_hit_groups, thr, Cn, _ptr(taxon), _ptr(cls),
                                            _ptr(nd), _ptr(tk), _ptr(hit_off), _ptr(hits), cap))
        out = dict(taxon=taxon, classified=cls, num_distinct=nd, total_kmers=tk)
        if with_hits:
            out["hit_offsets"] = hit_off
            out["hits"] = hits[:int(hit_off[R])]
            out["num_hits"] = np.diff(hit_off.astype(np.int64)).astype(np.int32)
        elif with_num_hits:
            out["num_hits"] = np.diff(hit_off.astype(np.int64)).astype(np.int32)
        return out

    def classify_hits(self, hit_offsets, hits, distinct=None, min_hit_groups=2, thresholds=(0.0,)):
        """Classifier.classify on caller-assembled hit lists (HIT_DTYPE array + offsets); distinct: uint8 per hit or None."""
        hit_offsets = _np(hit_offsets, np.uint64)
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        if distinct is not None:
            distinct = _np(distinct, np.uint8)
        R = hit_offsets.size - 1
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        taxon, cls = np.zeros((Cn, R), np.int32), np.zeros((Cn, R), np.uint8)
        nd, tk = np.zeros(R, np.int32), np.zeros(R, np.int32)
        _check(lib().slk_classify_hits(self.index.h, self.h, R, _ptr(hit_offsets), _ptr(hits), _ptr(distinct), min_hit_groups,
                                       thr, Cn, _ptr(taxon), _ptr(cls), _ptr(nd), _ptr(tk)))
        return dict(taxon=taxon, classified=cls, num_distinct=nd, total_kmers=tk)

    def classify_batch_device(self, d_bases, d_offsets, R, total_bases, d_out_taxon, d_out_classified,
                              d_out_num_distinct=None, d_out_total_kmers=None, d_out_num_hits=None, d_out_num_probes=None,
                              d_mate_bases=None, d_mate_offsets=None, total_mate_bases=0, min_hit_groups=2,
                              thresholds=(0.0,)):
        """All d_* are raw device addresses (ints), e.g. torch.Tensor.data_ptr(). Asynchronous on this stream."""
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        _check(lib().slk_classify_batch_device(self.index.h, self.h, d_bases, d_offsets, d_mate_bases,
                                               d_mate_offsets, R, total_bases, total_mate_bases, min_hit_groups, thr,
                                               Cn, d_out_taxon, d_out_classified, d_out_num_distinct,
                                               d_out_total_kmers, d_out_num_hits, d_out_num_probes))

    def scan_device(self, d_bases, d_offsets, R, d_span_keys, d_span_meta, d_span_count, d_mate_bases=None,
                    d_mate_offsets=None):
        _check(lib().slk_scan_device(self.index.h, self.h, d_bases, d_offsets, d_mate_bases, d_mate_offsets, R,
                                     d_span_keys, d_span_meta, d_span_count))

    def lookup_device(self, d_keys, n, d_out_taxa):
        _check(lib().slk_lookup_device(self.index.h, self.h, d_keys, n, d_out_taxa))

    def shard_step(self, emit=None, lookup=None, apply_lists=None, apply=None):
        """One pipeline step of the table-sharded mode (slk_shard_step_device): emit / apply_lists: ShardLists, lookup: ShardLookup,
        apply: ShardResults -- any may be None.  Asynchronous on this stream."""
        _check(lib().slk_shard_step_device(self.index.h, self.h, C.byref(emit) if emit is not None else None,
                                           C.byref(lookup) if lookup is not None else None,
                                           C.byref(apply_lists) if apply_lists is not None else None,
                                           C.byref(apply) if apply is not None else None))

    def classify_hits_device(self, d_offsets, R, d_span_meta, d_span_taxon, d_span_count, d_scratch, d_out_taxon,
                             d_out_classified, d_out_num_distinct=None, d_out_total_kmers=None, d_out_num_hits=None,
                             d_mate_offsets=None, min_hit_groups=2, thresholds=(0.0,)):
        Cn = len(thresholds)
        thr = (C.c_double * Cn)(*thresholds)
        _check(lib().slk_classify_hits_device(self.index.h, self.h, d_offsets, d_mate_offsets, R, d_span_meta,
                                              d_span_taxon, d_span_count, d_scratch, min_hit_groups, thr, Cn,
                                              d_out_taxon, d_out_classified, d_out_num_distinct, d_out_total_kmers,
                                              d_out_num_hits))

    def last_stage_ms(self):
        out = (C.c_float * 3)()
        _check(lib().slk_stream_last_stage_ms(self.h, out))
        return list(out)

    def close(self):
        """Destroys the engine stream -- unless torch still holds a view of it (external_stream): then the handle is parked in
        capi._deferred_streams, alive, and release_deferred_streams() destroys it once the views are gone."""
        if getattr(self, "h", None):
            if self.views_alive():
                _deferred_streams.append((self.h, list(self._views), self.index))
                self.h = None
                return
            lib().slk_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def release_deferred_streams():
    """Destroys the parked streams whose torch views have all died; -> how many are still parked."""
    keep = []
    for h, views, index in _deferred_streams:
        if any(r() is not None for r in views):
            keep.append((h, views, index))
        else:
            lib().slk_stream_destroy(h)
    _deferred_streams[:] = keep
    return len(keep)
