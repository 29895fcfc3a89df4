"""TEST INFRASTRUCTURE -- ctypes binding of oracle/_ref/libunc_oracle.so (the plain-C restatement,
oracle/unc_oracle.c).  Import only from tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); the product package never touches it."""
import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "_ref" / "libunc_oracle.so"

O_EVENT = np.dtype([("mean", "<f4"), ("stdv", "<f4"), ("start", "<u4"), ("length", "<u4")])
O_PATH = np.dtype([("fm_start", "<u8"), ("fm_end", "<u8"), ("event_moves", "<u4"), ("seed_prob", "<f4"),
                   ("kmer", "<u2"), ("length", "u1"), ("consec_stays", "u1"), ("sa_checked", "u1"),
                   ("pad", "u1", 3), ("prob_sums", "<f4", 23)], align=True)
O_CLUSTER = np.dtype([("ref_st", "<u8"), ("ref_en_start", "<u8"), ("ref_en_end", "<u8"),
                      ("evt_st", "<u4"), ("evt_en", "<u4"), ("total_len", "<u4"), ("pad", "<u4")])
O_HIT = np.dtype([("mapped", "<i4"), ("fwd", "<i4"), ("rid", "<i4"), ("matches", "<u4"),
                  ("rd_st", "<u8"), ("rd_en", "<u8"), ("rd_len", "<u8"),
                  ("rf_st", "<u8"), ("rf_en", "<u8"), ("rf_len", "<u8"),
                  ("n_events", "<u4"), ("event_i", "<u4"), ("mean_event_len", "<f4"), ("notes", "<u4"),
                  ("n_nbr", "<u8"), ("n_sa", "<u8"), ("n_lf", "<u8"), ("cluster", O_CLUSTER)])


class Params(C.Structure):
    _fields_ = [("seed_len", C.c_uint32), ("min_rep_len", C.c_uint32), ("max_rep_copy", C.c_uint32),
                ("max_paths", C.c_uint32), ("max_consec_stay", C.c_uint32), ("max_events", C.c_uint32),
                ("max_stay_frac", C.c_float), ("min_seed_prob", C.c_float),
                ("window_length1", C.c_uint32), ("window_length2", C.c_uint32),
                ("threshold1", C.c_float), ("threshold2", C.c_float), ("peak_height", C.c_float),
                ("min_mean", C.c_float), ("max_mean", C.c_float),
                ("min_map_len", C.c_uint32), ("min_mean_conf", C.c_float), ("min_top_conf", C.c_float),
                ("bp_per_sec", C.c_float), ("sample_rate", C.c_float)]


RT_TAP = np.dtype([("det_t", "<u4"), ("det_total_events", "<u4"), ("det_len_sum", "<f4"), ("norm_n", "<u4"), ("norm_wr", "<u4"),
                   ("prof_n", "<u4"), ("prof_to_mask", "<u4"), ("prof_queued", "<u4"), ("norm_mean", "<f8"), ("norm_varsum", "<f8"),
                   ("prof_mean", "<f8"), ("prof_varsum", "<f8"), ("prof_queue", "<f4", (28,))], align=True)

_lib = None


def available():
    return LIB_PATH.exists()


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(str(LIB_PATH))
        vp, u32, u64, f32p, u32p = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.unc_o_params_default.argtypes = [C.POINTER(Params)]
        L.unc_o_index_load.argtypes = [C.c_char_p, C.c_char_p]; L.unc_o_index_load.restype = vp
        L.unc_o_index_free.argtypes = [vp]
        L.unc_o_index_size.argtypes = [vp]; L.unc_o_index_size.restype = u64
        L.unc_o_index_ref_name.argtypes = [vp, C.c_int]; L.unc_o_index_ref_name.restype = C.c_char_p
        L.unc_o_index_kmer_ranges.argtypes = [vp, vp]
        L.unc_o_index_thresholds.argtypes = [vp, vp]
        L.unc_o_index_get_neighbor.argtypes = [vp, u64, u64, C.c_int, C.POINTER(u64), C.POINTER(u64)]
        L.unc_o_index_sa.argtypes = [vp, u64]; L.unc_o_index_sa.restype = u64
        L.unc_o_calibrate.argtypes = [vp, u64, C.c_float, C.c_float, C.c_float, vp]
        L.unc_o_detect_events.argtypes = [C.POINTER(Params), vp, u32, vp, u32, f32p, u32p]; L.unc_o_detect_events.restype = u32
        L.unc_o_model_tables.argtypes = [vp, vp, vp, f32p, f32p]
        L.unc_o_normalize.argtypes = [vp, u32, vp, f32p, f32p]
        L.unc_o_match_probs.argtypes = [C.c_float, vp]
        L.unc_o_mapper_new.argtypes = [vp, C.POINTER(Params)]; L.unc_o_mapper_new.restype = vp
        L.unc_o_mapper_free.argtypes = [vp]
        L.unc_o_map_read.argtypes = [vp, vp, u32, vp]
        L.unc_o_map_batch.argtypes = [vp, C.POINTER(Params), C.c_int, u32, vp, vp, vp]; L.unc_o_map_batch.restype = C.c_double
        L.unc_o_chunk_read.argtypes = [vp, vp, u32, u32, vp, u32p]
        L.unc_o_set_max_chunks.argtypes = [vp, u32]
        L.unc_o_trace_begin.argtypes = [vp, vp, u32]
        L.unc_o_trace_step.argtypes = [vp]
        L.unc_o_trace_paths.argtypes = [vp, vp, u32]; L.unc_o_trace_paths.restype = u32
        L.unc_o_trace_clusters.argtypes = [vp, vp, u32, vp, f32p, u32p]; L.unc_o_trace_clusters.restype = u32
        L.unc_o_trace_event_i.argtypes = [vp]; L.unc_o_trace_event_i.restype = u32
        L.unc_o_trace_finish.argtypes = [vp, vp]
        L.unc_o_stats.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), u32p, C.POINTER(u64)]
        _lib = L
    return _lib


def default_params():
    p = Params()
    lib().unc_o_params_default(C.byref(p))
    return p


def hit_paf_cols(h, ref_names):
    """PAF columns 2-12 (read_buffer.cpp:92-118) from an O_HIT record."""
    if not h["mapped"]:
        return (int(h["rd_len"]), "*")
    name = ref_names[int(h["rid"])] if h["rid"] >= 0 else ""
    return (int(h["rd_len"]), int(h["rd_st"]), int(h["rd_en"]), "+" if h["fwd"] else "-", name,
            int(h["rf_len"]), int(h["rf_st"]), int(h["rf_en"]), int(h["matches"]),
            int(h["rf_en"] - h["rf_st"] + 1), 255)


class Index:
    def __init__(self, prefix, preset="default"):
        self.h = lib().unc_o_index_load(str(prefix).encode(), preset.encode())
        if not self.h:
            raise RuntimeError(f"oracle: failed to load index {prefix}")
        self.size = lib().unc_o_index_size(self.h)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:      # module globals are gone at interpreter shutdown
            lib().unc_o_index_free(self.h)
            self.h = None

    def ref_name(self, rid):
        return lib().unc_o_index_ref_name(self.h, rid).decode()

    def ref_names(self):
        out, i = [], 0
        while True:
            n = self.ref_name(i)
            if not n:
                return out
            out.append(n)
            i += 1

    def kmer_ranges(self):
        out = np.empty((1024, 2), dtype=np.uint64)
        lib().unc_o_index_kmer_ranges(self.h, out.ctypes.data)
        return out

    def thresholds(self):
        out = np.empty(64, dtype=np.float32)
        lib().unc_o_index_thresholds(self.h, out.ctypes.data)
        return out

    def get_neighbor(self, s, e, b):
        os_, oe = C.c_uint64(), C.c_uint64()
        lib().unc_o_index_get_neighbor(self.h, s, e, b, C.byref(os_), C.byref(oe))
        return os_.value, oe.value

    def sa(self, k):
        return lib().unc_o_index_sa(self.h, k)


def calibrate(raw_i16, rng, offset, digitisation):
    raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
    out = np.empty(raw.size, dtype=np.float32)
    lib().unc_o_calibrate(raw.ctypes.data, raw.size, rng, offset, digitisation, out.ctypes.data)
    return out


def detect_events(signal_f32, params=None):
    p = params or default_params()
    sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
    out = np.zeros(sig.size // 2 + 16, dtype=O_EVENT)
    mel, tot = C.c_float(), C.c_uint32()
    n = lib().unc_o_detect_events(C.byref(p), sig.ctypes.data, sig.size, out.ctypes.data, out.size, C.byref(mel), C.byref(tot))
    return out[:n].copy(), mel.value, tot.value


def model_tables():
    a, b, c = (np.empty(1024, dtype=np.float32) for _ in range(3))
    mm, ms = C.c_float(), C.c_float()
    lib().unc_o_model_tables(a.ctypes.data, b.ctypes.data, c.ctypes.data, C.byref(mm), C.byref(ms))
    return a, b, c, mm.value, ms.value


def normalize(means_f32):
    m = np.ascontiguousarray(means_f32, dtype=np.float32)
    out = np.empty(m.size, dtype=np.float32)
    sc, sh = C.c_float(), C.c_float()
    lib().unc_o_normalize(m.ctypes.data, m.size, out.ctypes.data, C.byref(sc), C.byref(sh))
    return out, sc.value, sh.value


def match_probs(level):
    out = np.empty(1024, dtype=np.float32)
    lib().unc_o_match_probs(np.float32(level), out.ctypes.data)
    return out


class Mapper:
    def __init__(self, index, params=None):
        self.index = index
        self.params = params or default_params()
        self.h = lib().unc_o_mapper_new(index.h, C.byref(self.params))
        if not self.h:
            raise RuntimeError("oracle: unsupported params")

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:
            lib().unc_o_mapper_free(self.h)
            self.h = None

    def map_read(self, signal_f32):
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        hit = np.zeros(1, dtype=O_HIT)
        lib().unc_o_map_read(self.h, sig.ctypes.data, sig.size, hit.ctypes.data)
        return hit[0]

    def chunk_read(self, signal_f32, chunk_len=4000):
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        hit = np.zeros(1, dtype=O_HIT)
        used = C.c_uint32()
        lib().unc_o_chunk_read(self.h, sig.ctypes.data, sig.size, chunk_len, hit.ctypes.data, C.byref(used))
        return hit[0], used.value

    def rt_ended(self):
        """the ENDED flag of the read chunk_read mapped last"""
        f = lib().unc_o_rt_ended
        f.argtypes = [C.c_void_p]
        f.restype = C.c_int
        return bool(f(self.h))

    def set_max_chunks(self, n):
        lib().unc_o_set_max_chunks(self.h, n)

    def rt_tap(self):
        """(tap record, ring of 6000 floats): the state the chunked path carries between chunks below the PAF"""
        tap = np.zeros(1, dtype=RT_TAP)
        ring = np.zeros(6000, dtype=np.float32)
        f = lib().unc_o_rt_tap
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = None
        f(self.h, tap.ctypes.data, ring.ctypes.data)
        return tap[0], ring

    def stats(self):
        a, b, d = C.c_uint64(), C.c_uint64(), C.c_uint64()
        c = C.c_uint32()
        lib().unc_o_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(parents=a.value, children=b.value, max_children=c.value, seeds=d.value)

    def trace(self, signal_f32, max_clusters=1 << 16):
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        L = lib()
        L.unc_o_trace_begin(self.h, sig.ctypes.data, sig.size)
        mp = self.params.max_paths
        paths = np.zeros(mp, dtype=O_PATH)
        clus = np.zeros(max_clusters, dtype=O_CLUSTER)
        mm = np.zeros(1, dtype=O_CLUSTER)
        while True:
            done = L.unc_o_trace_step(self.h)
            n = L.unc_o_trace_paths(self.h, paths.ctypes.data, mp)
            ls, nl = C.c_float(), C.c_uint32()
            nc = L.unc_o_trace_clusters(self.h, clus.ctypes.data, max_clusters, mm.ctypes.data, C.byref(ls), C.byref(nl))
            yield bool(done), L.unc_o_trace_event_i(self.h), paths[:n].copy(), clus[:nc].copy(), mm[0].copy(), ls.value, nl.value
            if done:
                break

    def trace_finish(self):
        hit = np.zeros(1, dtype=O_HIT)
        lib().unc_o_trace_finish(self.h, hit.ctypes.data)
        return hit[0]


def map_batch(index, signals_f32, offsets_u64, n_threads=1, params=None):
    p = params or default_params()
    sig = np.ascontiguousarray(signals_f32, dtype=np.float32)
    off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
    hits = np.zeros(off.size - 1, dtype=O_HIT)
    secs = lib().unc_o_map_batch(index.h, C.byref(p), n_threads, off.size - 1, sig.ctypes.data, off.ctypes.data, hits.ctypes.data)
    return hits, secs
