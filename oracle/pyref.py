"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libunc_ref.so (the reference's own
hot-path object code behind oracle/ref_harness.h).  Import only from tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke()."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_ref" / "libunc_ref.so"


class RefHit(C.Structure):
    _fields_ = [("mapped", C.c_int32), ("fwd", C.c_int32),
                ("rd_st", C.c_uint64), ("rd_en", C.c_uint64), ("rd_len", C.c_uint64),
                ("rf_st", C.c_uint64), ("rf_en", C.c_uint64), ("rf_len", C.c_uint64),
                ("matches", C.c_uint32), ("n_events", C.c_uint32), ("event_i", C.c_uint32),
                ("mean_event_len", C.c_float),
                ("n_nbr", C.c_uint64), ("n_sa", C.c_uint64), ("n_lf", C.c_uint64),
                ("map_ms", C.c_double), ("rf_name", C.c_char * 128)]

    def paf_cols(self):
        """PAF columns 2-12 (read_buffer.cpp:92-118) as a tuple; name-independent."""
        if not self.mapped:
            return (int(self.rd_len), "*")
        return (int(self.rd_len), int(self.rd_st), int(self.rd_en), "+" if self.fwd else "-",
                self.rf_name.decode(), int(self.rf_len), int(self.rf_st), int(self.rf_en),
                int(self.matches), int(self.rf_en - self.rf_st + 1), 255)


REF_EVENT = np.dtype([("mean", "<f4"), ("stdv", "<f4"), ("start", "<u4"), ("length", "<u4")])
REF_PATH = np.dtype([("fm_start", "<u8"), ("fm_end", "<u8"), ("event_moves", "<u4"), ("seed_prob", "<f4"),
                     ("kmer", "<u2"), ("length", "u1"), ("consec_stays", "u1"), ("sa_checked", "u1"),
                     ("pad", "u1", 3), ("prob_sums", "<f4", 23)], align=True)
REF_CLUSTER = np.dtype([("ref_st", "<u8"), ("ref_en_start", "<u8"), ("ref_en_end", "<u8"),
                        ("evt_st", "<u4"), ("evt_en", "<u4"), ("total_len", "<u4"), ("pad", "<u4")])

_lib = None
_prefix = None


def available():
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(LIB_PATH))
        L.ref_init.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
        L.ref_mapper_new.restype = C.c_void_p
        L.ref_mapper_free.argtypes = [C.c_void_p]
        L.ref_calibrate.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.ref_map_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(RefHit)]
        L.ref_map_batch.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_map_batch.restype = C.c_double
        L.ref_chunk_read.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(RefHit), C.POINTER(C.c_uint32)]
        L.ref_set_max_chunks.argtypes = [C.c_uint32]
        L.ref_set_sort_mode.argtypes = [C.c_int]
        L.ref_sort_stats.argtypes = [C.c_void_p, C.c_int]
        L.ref_sort_selftest.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.ref_events.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.ref_events.restype = C.c_uint32
        L.ref_norm_levels.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ref_match_probs.argtypes = [C.c_float, C.c_void_p]
        L.ref_model_tables.argtypes = [C.c_void_p] * 3 + [C.POINTER(C.c_float)] * 2
        L.ref_kmer_ranges.argtypes = [C.c_void_p]
        L.ref_thresholds.argtypes = [C.c_void_p]
        L.ref_get_neighbor.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ref_sa.argtypes = [C.c_uint64]
        L.ref_sa.restype = C.c_uint64
        L.ref_fm_size.restype = C.c_uint64
        L.ref_trace_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ref_trace_step.argtypes = [C.c_void_p]
        L.ref_trace_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ref_trace_paths.restype = C.c_uint32
        L.ref_trace_clusters.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        L.ref_trace_clusters.restype = C.c_uint32
        L.ref_trace_event_i.argtypes = [C.c_void_p]
        L.ref_trace_event_i.restype = C.c_uint32
        L.ref_trace_finish.argtypes = [C.c_void_p, C.POINTER(RefHit)]
        _lib = L
    return _lib


def init(prefix, preset="default", max_events=0):
    """One index per process (the reference keeps it in statics: mapper.hpp:80-85)."""
    global _prefix
    prefix = str(prefix)
    if _prefix is not None:
        if _prefix != prefix:
            raise RuntimeError(f"reference statics already loaded with {_prefix}")
        return
    if lib().ref_init(prefix.encode(), preset.encode(), max_events) != 0:
        raise RuntimeError("ref_init failed")
    _prefix = prefix


def set_max_paths(n):
    lib().ref_set_max_paths.argtypes = [C.c_uint32]
    lib().ref_set_max_paths(n)


SORT_STABLE, SORT_PDQ_RESTATED, SORT_REVERSED_TIES = 0, 1, 2


def set_sort_mode(mode):
    """order of children that tie on (fm_range_, seed_prob_) at mapper.cpp:531 (oracle/shim/pdqsort.h), for every sort from now on"""
    lib().ref_set_sort_mode(int(mode))


def sort_stats(reset=False):
    """(sorts = events with children, sorts that had a tied adjacent pair, tied adjacent pairs) since the last reset"""
    out = np.zeros(3, dtype=np.uint64)
    lib().ref_sort_stats(out.ctypes.data, 1 if reset else 0)
    return int(out[0]), int(out[1]), int(out[2])


def sort_selftest(n, seed, key_range, shape=0):
    return int(lib().ref_sort_selftest(n, seed, key_range, shape))


def set_params(p):
    """Mapper::PRMS from a pyoracle.Params / capi.Params (same field names), for Mappers constructed afterwards"""
    f = lib().ref_set_params
    f.argtypes = [C.c_uint32] * 5 + [C.c_float] * 7 + [C.c_uint32, C.c_float, C.c_float]
    f(p.min_rep_len, p.max_rep_copy, p.max_paths, p.max_consec_stay, p.max_events, p.max_stay_frac, p.min_seed_prob,
      p.threshold1, p.threshold2, p.peak_height, p.min_mean, p.max_mean, p.min_map_len, p.min_mean_conf, p.min_top_conf)


def calibrate(raw_i16, rng, offset, digitisation):
    raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
    out = np.empty(raw.size, dtype=np.float32)
    lib().ref_calibrate(raw.ctypes.data, raw.size, rng, offset, digitisation, out.ctypes.data)
    return out


class Mapper:
    def __init__(self):
        self.h = lib().ref_mapper_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_mapper_free(self.h)
            self.h = None

    def map_read(self, signal_f32):
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        hit = RefHit()
        lib().ref_map_read(self.h, sig.ctypes.data, sig.size, C.byref(hit))
        return hit

    def chunk_read(self, signal_f32, chunk_len=4000, number=0):
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        hit, used = RefHit(), C.c_uint32()
        lib().ref_chunk_read(self.h, sig.ctypes.data, sig.size, chunk_len, number, C.byref(hit), C.byref(used))
        return hit, used.value

    def rt_tap(self):
        from oracle.pyoracle import RT_TAP
        tap = np.zeros(1, dtype=RT_TAP)
        ring = np.zeros(6000, dtype=np.float32)
        f = lib().ref_rt_tap
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        f.restype = None
        f(self.h, tap.ctypes.data, ring.ctypes.data, 6000)
        return tap[0], ring

    def trace(self, signal_f32, max_paths=10000, max_clusters=1 << 16):
        """Generator: after each map_next() yields (done, event_i, paths, clusters, max_map, len_sum, n_lens)."""
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        L = lib()
        L.ref_trace_begin(self.h, sig.ctypes.data, sig.size)
        paths = np.zeros(max_paths, dtype=REF_PATH)
        clus = np.zeros(max_clusters, dtype=REF_CLUSTER)
        mm = np.zeros(1, dtype=REF_CLUSTER)
        while True:
            done = L.ref_trace_step(self.h)
            n = L.ref_trace_paths(self.h, paths.ctypes.data, max_paths)
            ls, nl = C.c_float(), C.c_uint32()
            nc = L.ref_trace_clusters(self.h, clus.ctypes.data, max_clusters, mm.ctypes.data, C.byref(ls), C.byref(nl))
            yield bool(done), L.ref_trace_event_i(self.h), paths[:n].copy(), clus[:nc].copy(), mm[0].copy(), ls.value, nl.value
            if done:
                break

    def trace_finish(self):
        hit = RefHit()
        lib().ref_trace_finish(self.h, C.byref(hit))
        return hit


def map_batch(signals_f32, offsets_u64, n_threads, pool=False):
    """tight new_read -> map_read loop on n_threads (SURVEY 8d "B1"); pool=True: through the as-shipped MapPool hand-shake
    with its 10 ms polling sleeps ("B2")"""
    sig = np.ascontiguousarray(signals_f32, dtype=np.float32)
    off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
    n = off.size - 1
    hits = (RefHit * n)()
    L = lib()
    fn = L.ref_map_batch_pool if pool else L.ref_map_batch
    fn.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    fn.restype = C.c_double
    secs = fn(n_threads, n, sig.ctypes.data, off.ctypes.data, C.cast(hits, C.c_void_p))
    return list(hits), secs


def self_align(prefix, sample_dist):
    """-> list of uint64 arrays (the reference's std::vector<std::vector<u64>>)."""
    L = lib()
    L.ref_self_align.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.ref_self_align.restype = C.c_uint64
    n = C.c_uint64()
    tot = L.ref_self_align(str(prefix).encode(), sample_dist, None, 0, None, 0, C.byref(n))
    lens = np.empty(tot, dtype=np.uint64)
    offs = np.empty(n.value + 1, dtype=np.uint64)
    L.ref_self_align(str(prefix).encode(), sample_dist, lens.ctypes.data, tot, offs.ctypes.data, n.value + 1, C.byref(n))
    return [lens[int(offs[i]):int(offs[i + 1])] for i in range(n.value)]


def events(signal_f32):
    sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
    out = np.zeros(sig.size // 2 + 16, dtype=REF_EVENT)
    mel, tot = C.c_float(), C.c_uint32()
    n = lib().ref_events(sig.ctypes.data, sig.size, out.ctypes.data, out.size, C.byref(mel), C.byref(tot))
    return out[:n].copy(), mel.value, tot.value


def norm_levels(means_f32):
    m = np.ascontiguousarray(means_f32, dtype=np.float32)
    out = np.empty(m.size, dtype=np.float32)
    sc, sh = C.c_float(), C.c_float()
    lib().ref_norm_levels(m.ctypes.data, m.size, out.ctypes.data, C.byref(sc), C.byref(sh))
    return out, sc.value, sh.value


def match_probs(level):
    out = np.empty(1024, dtype=np.float32)
    lib().ref_match_probs(np.float32(level), out.ctypes.data)
    return out


def model_tables():
    a, b, c = (np.empty(1024, dtype=np.float32) for _ in range(3))
    mm, ms = C.c_float(), C.c_float()
    lib().ref_model_tables(a.ctypes.data, b.ctypes.data, c.ctypes.data, C.byref(mm), C.byref(ms))
    return a, b, c, mm.value, ms.value


def kmer_ranges():
    out = np.empty((1024, 2), dtype=np.uint64)
    lib().ref_kmer_ranges(out.ctypes.data)
    return out


def thresholds():
    out = np.empty(64, dtype=np.float32)
    lib().ref_thresholds(out.ctypes.data)
    return out


def get_neighbor(s, e, b):
    os_, oe = C.c_uint64(), C.c_uint64()
    lib().ref_get_neighbor(s, e, b, C.byref(os_), C.byref(oe))
    return os_.value, oe.value


def sa(k):
    return lib().ref_sa(k)


def fm_size():
    return lib().ref_fm_size()
