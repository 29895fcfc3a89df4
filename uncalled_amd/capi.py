"""ctypes binding of the C ABI in include/uncalled_hip.h.

`load()` opens uncalled_amd/libuncalled_hip.so -- the gfx950 code object built by
`__graft_entry__.build()` -- and raises if it is missing: there is no CPU mapping path behind this
package.  (tests/ may pass an explicit library path to run the same kernel sources under the
lanesim CPU emulator; the package itself never does.)
"""
import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
DEFAULT_LIB = HERE / "libuncalled_hip.so"

UNC_OK = 0
UNC_ERR_OVERFLOW = -5


class Params(C.Structure):
    """unc_params_t: Mapper::PRMS + EventDetector/SeedTracker/ReadBuffer params (reference defaults)."""
    _fields_ = [("seed_len", C.c_uint32), ("min_rep_len", C.c_uint32), ("max_rep_copy", C.c_uint32),
                ("max_paths", C.c_uint32), ("max_consec_stay", C.c_uint32), ("max_events", C.c_uint32),
                ("max_stay_frac", C.c_float), ("min_seed_prob", C.c_float),
                ("window_length1", C.c_uint32), ("window_length2", C.c_uint32),
                ("threshold1", C.c_float), ("threshold2", C.c_float), ("peak_height", C.c_float),
                ("min_mean", C.c_float), ("max_mean", C.c_float),
                ("min_map_len", C.c_uint32), ("min_mean_conf", C.c_float), ("min_top_conf", C.c_float),
                ("bp_per_sec", C.c_float), ("sample_rate", C.c_float), ("chunk_time", C.c_float),
                ("max_chunks", C.c_uint32)]


class MapperOpts(C.Structure):
    _fields_ = [("n_slots", C.c_uint32), ("max_clusters", C.c_uint32), ("max_seed_paths", C.c_uint32),
                ("slice_events", C.c_uint32), ("n_waves", C.c_uint32), ("pool_chunks", C.c_uint32), ("sched_parts", C.c_uint32),
                ("events_reads_per_wave", C.c_uint32)]


CALIB = np.dtype([("range", "<f4"), ("offset", "<f4"), ("digitisation", "<f4")])
HIT = np.dtype([("mapped", "<i4"), ("fwd", "<i4"), ("rid", "<i4"), ("status", "<u4"),
                ("rd_st", "<u8"), ("rd_en", "<u8"), ("rd_len", "<u8"),
                ("rf_st", "<u8"), ("rf_en", "<u8"), ("rf_len", "<u8"),
                ("matches", "<u4"), ("n_events", "<u4"), ("event_i", "<u4"), ("mean_event_len", "<f4"),
                ("n_nbr", "<u8"), ("n_sa", "<u8"), ("n_lf", "<u8"),
                ("cl_ref_st", "<u8"), ("cl_ref_en_start", "<u8"), ("cl_ref_en_end", "<u8"),
                ("cl_evt_st", "<u4"), ("cl_evt_en", "<u4"), ("cl_total_len", "<u4"), ("map_ms", "<f4"),
                ("notes", "<u4"), ("pad_", "<u4")])
NOTE_PATHS_FULL, NOTE_FLAGS_LEFT = 1, 2      # unc_hit_t::notes (include/uncalled_hip.h)
ORDER_INDEPENDENT, ORDER_T1 = 0, 1           # unc_mapper_set_read_order
# every field of a hit but the timing one: what two runs over the same reads must agree on
RESULT_FIELDS = tuple(n for n in HIT.names if n != "map_ms")


def sort_pairs_device(keys_ptr, vals_ptr, tmp_keys_ptr, tmp_vals_ptr, n, key_bits=64, iota=False, device=0, stream=None, lib=None):
    """unc_sort_pairs_u64 on device pointers (the index builders pass torch tensors' data_ptr): stable LSD radix sort of n (key, value)
    pairs of 64 bits by the low key_bits of the key, in place; iota: the values are 0 .. n - 1, so `vals` returns the permutation."""
    L = lib or load()
    _check(L, L.unc_sort_pairs_u64(int(device), int(n), keys_ptr, vals_ptr, tmp_keys_ptr, tmp_vals_ptr, int(key_bits), 1 if iota else 0, stream))


def sort_pairs(keys, vals=None, key_bits=64, lib=None, device=0):
    """The same sort on host arrays (tests): staged in page-locked host memory, which the kernels read and write directly.
    -> (sorted keys uint64, values uint64); vals None = the sorting permutation."""
    L = lib or load()
    k = np.ascontiguousarray(keys, dtype=np.uint64)
    n = int(k.size)
    if n == 0:
        return k.copy(), np.zeros(0, np.uint64)
    buf = L.unc_host_alloc(4 * n * 8)
    if not buf:
        raise UncalledHipError("unc_host_alloc failed")
    try:
        a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64)), shape=(4 * n,))
        a[:n] = k
        if vals is not None:
            a[n:2 * n] = np.ascontiguousarray(vals, dtype=np.uint64)
        sort_pairs_device(buf, buf + 8 * n, buf + 16 * n, buf + 24 * n, n, key_bits, vals is None, device, None, L)
        return a[:n].copy(), a[n:2 * n].copy()
    finally:
        L.unc_host_free(buf)


def build_suffix_array(codes_u8, device=0, lib=None):
    """unc_build_suffix_array: the suffix array (int64) of a text of n < 2^31 symbols (codes 0..3), built on the device without torch."""
    L = lib or load()
    t = np.ascontiguousarray(codes_u8, dtype=np.uint8)
    sa = np.empty(t.size, dtype=np.int64)
    _check(L, L.unc_build_suffix_array(int(device), t.ctypes.data, int(t.size), sa.ctypes.data))
    return sa


def hits_digest(hits):
    """sha256 over the result fields of a hit array (map_ms, a wall-clock measurement, zeroed)."""
    import hashlib
    h = hits.copy()
    h["map_ms"] = 0
    return hashlib.sha256(h.tobytes()).hexdigest()


EVT_INFO = np.dtype([("n_events", "<u4"), ("total_events", "<u4"), ("len_sum", "<f4"), ("scale", "<f4"),
                     ("shift", "<f4"), ("pad", "<u4")])
PATH = np.dtype([("fm_start", "<u8"), ("fm_end", "<u8"), ("event_moves", "<u4"), ("seed_prob", "<f4"),
                 ("kmer", "<u2"), ("length", "u1"), ("consec_stays", "u1"), ("sa_checked", "u1"),
                 ("pad", "u1", 3), ("prob_sums", "<f4", 23)], align=True)
CLUSTER = np.dtype([("ref_st", "<u8"), ("ref_en_start", "<u8"), ("ref_en_end", "<u8"),
                    ("evt_st", "<u4"), ("evt_en", "<u4"), ("total_len", "<u4"), ("pad", "<u4")])

RT_CHUNK = np.dtype([("channel", "<u4"), ("read_number", "<u4"), ("flags", "<u4"), ("n_samples", "<u4"), ("offset", "<u8"),
                     ("calib", CALIB), ("pad", "<u4")])
RT_RESULT = np.dtype([("state", "<i4"), ("ended", "<i4"), ("hit", HIT)])
RT_FIRST, RT_LAST = 1, 2
RT_MAPPING, RT_MAPPED, RT_FAILED, RT_IGNORED = 0, 1, 2, 3

_libs = {}


class UncalledHipError(RuntimeError):
    pass


def load(path=None):
    """Open the shared library (default: the in-tree gfx950 build) and declare its prototypes."""
    path = Path(path) if path else DEFAULT_LIB
    key = str(path)
    if key in _libs:
        return _libs[key]
    if not path.exists():
        raise UncalledHipError(
            f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "uncalled_amd has no CPU fallback.")
    L = C.CDLL(str(path.resolve()))
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
    L.unc_last_error.restype = C.c_char_p
    L.unc_version.restype = C.c_char_p
    L.unc_calib_traffic.argtypes = [C.c_int, u64, C.c_int]
    L.unc_host_alloc.argtypes = [u64]; L.unc_host_alloc.restype = vp
    L.unc_host_free.argtypes = [vp]; L.unc_host_free.restype = None
    L.unc_params_default.argtypes = [C.POINTER(Params)]
    L.unc_index_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]
    L.unc_index_free.argtypes = [vp]
    L.unc_index_size.argtypes = [vp]; L.unc_index_size.restype = u64
    L.unc_index_n_seqs.argtypes = [vp]; L.unc_index_n_seqs.restype = i32
    L.unc_index_seq_name.argtypes = [vp, i32]; L.unc_index_seq_name.restype = C.c_char_p
    L.unc_index_seq_len.argtypes = [vp, i32]; L.unc_index_seq_len.restype = u64
    L.unc_index_translate_loc.argtypes = [vp, u64, C.POINTER(i32), C.POINTER(u64)]; L.unc_index_translate_loc.restype = u64
    L.unc_index_device_bytes.argtypes = [vp]; L.unc_index_device_bytes.restype = u64
    L.unc_index_kmer_ranges.argtypes = [vp, vp]
    L.unc_index_thresholds.argtypes = [vp, vp]
    L.unc_index_model_tables.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.unc_fm_get_neighbor.argtypes = [vp, u32, vp, vp, vp, vp, vp]
    L.unc_fm_sa.argtypes = [vp, u32, vp, vp]
    L.unc_match_probs.argtypes = [vp, u32, vp, vp]
    L.unc_self_align.argtypes = [vp, C.c_char_p, u32, u32, vp, vp, u64, C.POINTER(u64)]
    L.unc_mapper_create.argtypes = [vp, C.POINTER(Params), C.POINTER(MapperOpts), C.POINTER(vp)]
    L.unc_mapper_free.argtypes = [vp]
    L.unc_mapper_device_bytes.argtypes = [vp]; L.unc_mapper_device_bytes.restype = u64
    L.unc_map_batch.argtypes = [vp, u32, vp, vp, vp, C.c_int, vp, vp]
    if hasattr(L, "unc_build_suffix_array"):
        L.unc_build_suffix_array.argtypes = [C.c_int, vp, u64, vp]
    if hasattr(L, "unc_mapper_last_window"):
        L.unc_mapper_last_window.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "unc_map_batch_begin"):
        L.unc_map_batch_begin.argtypes = [vp, u32, vp, vp, vp, C.c_int, vp]
        L.unc_map_batch_end.argtypes = [vp, vp]
    L.unc_mapper_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.unc_mapper_last_phase_cycles.argtypes = [vp, vp]
    L.unc_mapper_last_read_cycles.argtypes = [vp, u32, vp]
    L.unc_mapper_last_remap.argtypes = [vp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.unc_mapper_last_remap.restype = None
    if hasattr(L, "unc_mapper_set_read_order"):      # (dev tools load older builds of the library for A/B runs)
        L.unc_mapper_set_read_order.argtypes = [vp, C.c_int]
        L.unc_mapper_last_carry_over.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_float)]
    L.unc_mapper_kernel_info.argtypes = [vp, vp]
    L.unc_mapper_kernel_info.restype = i32
    L.unc_mapper_geometry.argtypes = [vp, vp]
    L.unc_mapper_geometry.restype = None
    if hasattr(L, "unc_sort_pairs_u64"):
        L.unc_sort_pairs_u64.argtypes = [i32, u64, vp, vp, vp, vp, i32, i32, vp]
        L.unc_sort_pairs_u64.restype = C.c_int
    if hasattr(L, "unc_mapper_pool_usage"):           # (older builds under uncalled_amd/variants/ lack it)
        L.unc_mapper_pool_usage.argtypes = [vp, vp]
        L.unc_mapper_pool_usage.restype = C.c_int
    L.unc_mapper_set_profile.argtypes = [vp, C.c_int]
    L.unc_mapper_set_profile.restype = None
    L.unc_mapper_last_wave_busy.argtypes = [vp]
    L.unc_mapper_last_wave_busy.restype = C.c_double
    L.unc_detect_events.argtypes = [vp, u32, vp, vp, vp, vp, u64, vp, vp]
    L.unc_rt_create.argtypes = [vp, C.POINTER(Params), u32, C.POINTER(vp)]
    L.unc_rt_free.argtypes = [vp]
    L.unc_rt_device_bytes.argtypes = [vp]; L.unc_rt_device_bytes.restype = u64
    L.unc_rt_process_chunks.argtypes = [vp, u32, vp, vp, C.c_int, vp, vp]
    L.unc_rt_process_chunks_f32.argtypes = [vp, u32, vp, vp, C.c_int, vp, vp]
    L.unc_rt_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if hasattr(L, "unc_rt_tap_channel"):
        L.unc_rt_tap_channel.argtypes = [vp, u32, vp, vp]
    L.unc_trace_begin.argtypes = [vp, vp, u32, vp]
    L.unc_trace_step.argtypes = [vp, u32, C.POINTER(C.c_int)]
    L.unc_trace_paths.argtypes = [vp, vp, u32, C.POINTER(u32)]
    L.unc_trace_clusters.argtypes = [vp, vp, u32, C.POINTER(u32), vp, C.POINTER(C.c_float), C.POINTER(u32)]
    L.unc_trace_finish.argtypes = [vp, vp]
    _libs[key] = L
    return L


def _check(L, rc, allow=()):
    if rc != UNC_OK and rc not in allow:
        raise UncalledHipError(f"uncalled_hip error {rc}: {L.unc_last_error().decode()}")
    return rc


def default_params(lib=None):
    p = Params()
    (lib or load()).unc_params_default(C.byref(p))
    return p


class Index:
    """Device-resident FM index + thresholds + pore model (Mapper::load_static, mapper.cpp:109-159)."""

    def __init__(self, prefix, preset="default", device=0, lib=None):
        self.L = lib or load()
        h = C.c_void_p()
        _check(self.L, self.L.unc_index_load(str(prefix).encode(), preset.encode(), device, C.byref(h)))
        self.h = h
        self.size = self.L.unc_index_size(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.unc_index_free(self.h)
            self.h = None

    __del__ = close

    def seq_names(self):
        return [self.L.unc_index_seq_name(self.h, i).decode() for i in range(self.L.unc_index_n_seqs(self.h))]

    def seq_len(self, rid):
        return self.L.unc_index_seq_len(self.h, rid)

    def device_bytes(self):
        return self.L.unc_index_device_bytes(self.h)

    def kmer_ranges(self):
        out = np.empty((1024, 2), dtype=np.uint64)
        self.L.unc_index_kmer_ranges(self.h, out.ctypes.data)
        return out

    def thresholds(self):
        out = np.empty(64, dtype=np.float32)
        self.L.unc_index_thresholds(self.h, out.ctypes.data)
        return out

    def model_tables(self):
        a, b, c = (np.empty(1024, dtype=np.float32) for _ in range(3))
        mm, ms = C.c_float(), C.c_float()
        self.L.unc_index_model_tables(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data, C.byref(mm), C.byref(ms))
        return a, b, c, mm.value, ms.value

    def get_neighbor(self, starts, ends, bases):
        s = np.ascontiguousarray(starts, dtype=np.uint64)
        e = np.ascontiguousarray(ends, dtype=np.uint64)
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        os_, oe = np.empty_like(s), np.empty_like(s)
        _check(self.L, self.L.unc_fm_get_neighbor(self.h, s.size, s.ctypes.data, e.ctypes.data, b.ctypes.data,
                                                  os_.ctypes.data, oe.ctypes.data))
        return os_, oe

    def sa(self, rows):
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.empty_like(r)
        _check(self.L, self.L.unc_fm_sa(self.h, r.size, r.ctypes.data, out.ctypes.data))
        return out

    def self_align(self, prefix, sample_dist, cap=128):
        """self_align(bwa_prefix, sample_dist) of the reference: (lens uint64[n, cap], full_len uint32[n])."""
        n = C.c_uint64()
        _check(self.L, self.L.unc_self_align(self.h, str(prefix).encode(), sample_dist, cap, None, None, 0, C.byref(n)))
        lens = np.zeros((n.value, cap), dtype=np.uint64)
        full = np.zeros(n.value, dtype=np.uint32)
        _check(self.L, self.L.unc_self_align(self.h, str(prefix).encode(), sample_dist, cap, lens.ctypes.data, full.ctypes.data,
                                             n.value, C.byref(n)))
        return lens, full

    def match_probs(self, levels):
        lv = np.ascontiguousarray(levels, dtype=np.float32)
        out = np.empty((lv.size, 1024), dtype=np.float32)
        _check(self.L, self.L.unc_match_probs(self.h, lv.size, lv.ctypes.data, out.ctypes.data))
        return out


def make_calib(n, rng, offset, digitisation):
    c = np.empty(n, dtype=CALIB)
    c["range"], c["offset"], c["digitisation"] = rng, offset, digitisation
    return c


def hit_paf_cols(h, names):
    """PAF columns 2-12 (Paf::print_paf, read_buffer.cpp:92-118) of one HIT record."""
    if not h["mapped"]:
        return (int(h["rd_len"]), "*")
    name = names[int(h["rid"])] if h["rid"] >= 0 else ""
    return (int(h["rd_len"]), int(h["rd_st"]), int(h["rd_en"]), "+" if h["fwd"] else "-", name,
            int(h["rf_len"]), int(h["rf_st"]), int(h["rf_en"]), int(h["matches"]),
            int(h["rf_en"] - h["rf_st"] + 1), 255)


class Mapper:
    """Batch mapper: N x (Mapper::new_read + Mapper::map_read) on the GPU (mapper.cpp:188-207)."""

    def __init__(self, index, params=None, n_slots=0, max_clusters=0, max_seed_paths=0, slice_events=0, n_waves=0, pool_chunks=0,
                 events_reads_per_wave=0, sched_parts=0):
        self.index = index
        self.L = index.L
        self.params = params or default_params(self.L)
        opts = MapperOpts(n_slots, max_clusters, max_seed_paths, slice_events, n_waves, pool_chunks, sched_parts, events_reads_per_wave)
        h = C.c_void_p()
        _check(self.L, self.L.unc_mapper_create(index.h, C.byref(self.params), C.byref(opts), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.unc_mapper_free(self.h)
            self.h = None

    __del__ = close

    def device_bytes(self):
        return self.L.unc_mapper_device_bytes(self.h)

    def map_batch(self, raw_i16, offsets_u64, calib, allow_overflow=False):
        """raw: host int16 array (all reads concatenated); returns HIT[n_reads]."""
        raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
        off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        cal = np.ascontiguousarray(calib, dtype=CALIB)
        n = off.size - 1
        hits = np.zeros(n, dtype=HIT)
        _check(self.L, self.L.unc_map_batch(self.h, n, raw.ctypes.data, off.ctypes.data, cal.ctypes.data, 0, None,
                                            hits.ctypes.data), allow=(UNC_ERR_OVERFLOW,) if allow_overflow else ())
        return hits

    def map_batch_device(self, raw_ptr, offsets_u64, calib, stream=None, allow_overflow=False):
        """raw_ptr: integer device address of the int16 samples already resident in HBM."""
        off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        cal = np.ascontiguousarray(calib, dtype=CALIB)
        n = off.size - 1
        hits = np.zeros(n, dtype=HIT)
        _check(self.L, self.L.unc_map_batch(self.h, n, C.c_void_p(raw_ptr), off.ctypes.data, cal.ctypes.data, 1,
                                            C.c_void_p(stream or 0), hits.ctypes.data),
               allow=(UNC_ERR_OVERFLOW,) if allow_overflow else ())
        return hits

    def begin_batch(self, raw, offsets_u64, calib, on_device=False, stream=None):
        """unc_map_batch_begin: stage the reads and launch the kernels (raw: a host int16 array, or with on_device the integer device address
        of the samples); returns at once.  end_batch() waits and returns HIT[n_reads].  One batch per mapper at a time."""
        off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        cal = np.ascontiguousarray(calib, dtype=CALIB)
        self._pending_n = off.size - 1
        if on_device:
            ptr = C.c_void_p(raw)
        else:
            self._pending_raw = np.ascontiguousarray(raw, dtype=np.int16)      # (must outlive the call: the copy to the device is asynchronous)
            ptr = C.c_void_p(self._pending_raw.ctypes.data)
        _check(self.L, self.L.unc_map_batch_begin(self.h, self._pending_n, ptr, off.ctypes.data, cal.ctypes.data, 1 if on_device else 0,
                                                  C.c_void_p(stream or 0)))

    def end_batch(self, allow_overflow=False):
        hits = np.zeros(self._pending_n, dtype=HIT)
        _check(self.L, self.L.unc_map_batch_end(self.h, hits.ctypes.data), allow=(UNC_ERR_OVERFLOW,) if allow_overflow else ())
        self._pending_raw = None
        return hits

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        self.L.unc_mapper_last_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def last_window(self):
        """(start, end) of the last batch's k_map in ms on the process-wide time axis (unc_mapper_last_window)"""
        a, b = C.c_double(), C.c_double()
        _check(self.L, self.L.unc_mapper_last_window(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_remap(self):
        n, ms = C.c_uint32(), C.c_float()
        self.L.unc_mapper_last_remap(self.h, C.byref(n), C.byref(ms))
        return int(n.value), float(ms.value)

    def set_read_order(self, order):
        """ORDER_INDEPENDENT (default: every read as a fresh Mapper maps it) or ORDER_T1 (`uncalled map -t 1`: one Mapper, reads in
        the order given, sources_added_ carried from read to read and batch to batch); see include/uncalled_hip.h"""
        _check(self.L, self.L.unc_mapper_set_read_order(self.h, int(order)))

    def last_carry_over(self):
        """ORDER_T1: (reads of the last batch mapped again because their predecessor left flags set, rounds, ms)"""
        n, r, ms = C.c_uint32(), C.c_uint32(), C.c_float()
        self.L.unc_mapper_last_carry_over(self.h, C.byref(n), C.byref(r), C.byref(ms))
        return int(n.value), int(r.value), float(ms.value)

    def geometry(self):
        out = np.zeros(5, dtype=np.uint32)
        self.L.unc_mapper_geometry(self.h, out.ctypes.data)
        return dict(zip(("n_waves", "n_slots", "slice_events", "pool_chunks", "max_clusters"), (int(x) for x in out)))

    def sched_parts(self):
        """Pairs of scheduler rings (one per XCD where the slots divide; 0: no time slicing)."""
        self.L.unc_mapper_sched_parts.restype = C.c_uint32
        self.L.unc_mapper_sched_parts.argtypes = [C.c_void_p]
        return int(self.L.unc_mapper_sched_parts(self.h))

    def pool_usage(self):
        """Seed-cluster node pool: chunks (192 KB each) held now, high-water mark of chunks out at once in the last batch / ever,
        times the library resized it (cut to four times the high-water mark when it holds more than eight times that, doubled when found dry; include/uncalled_hip.h)."""
        if not hasattr(self.L, "unc_mapper_pool_usage"):
            return None
        out = np.zeros(4, dtype=np.uint32)
        _check(self.L, self.L.unc_mapper_pool_usage(self.h, out.ctypes.data))
        d = dict(zip(("chunks", "high_water_last_batch", "high_water_ever", "resizes"), (int(x) for x in out)))
        d["gb"] = round(d["chunks"] * 192 * 1024 / 1e9, 2)
        return d

    def kernel_info(self):
        """Register / scratch / LDS figures of the k_map instantiation this mapper launches, read off the code object."""
        out = np.zeros(6, dtype=np.uint32)
        _check(self.L, self.L.unc_mapper_kernel_info(self.h, out.ctypes.data))
        d = dict(zip(("vgprs", "scratch_bytes_per_lane", "lds_bytes", "max_threads", "waves_per_cu", "rows32"), (int(x) for x in out)))
        d["rows32"] = bool(d["rows32"])
        return d

    def set_profile(self, on=True):
        self.L.unc_mapper_set_profile(self.h, 1 if on else 0)

    def last_wave_busy(self):
        return float(self.L.unc_mapper_last_wave_busy(self.h))

    def last_phase_cycles(self):
        out = np.zeros(12, dtype=np.uint64)
        self.L.unc_mapper_last_phase_cycles(self.h, out.ctypes.data)
        return dict(zip(("probs", "extend_rest", "sort", "walk", "sources", "sa", "add_seed", "rest", "e1_parents", "e2_fm", "e3_slots", "e4_children"), out.tolist()))

    def last_read_cycles(self, n_reads):
        """Per read of the last batch: [n_reads, 14] -- twelve phase counters, residence ticks, XCC_ID | HW_ID << 8 (profiling on)."""
        out = np.zeros((int(n_reads), 14), dtype=np.uint64)
        _check(self.L, self.L.unc_mapper_last_read_cycles(self.h, int(n_reads), out.ctypes.data))
        return out

    def detect_events(self, raw_i16, offsets_u64, calib):
        raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
        off = np.ascontiguousarray(offsets_u64, dtype=np.uint64)
        cal = np.ascontiguousarray(calib, dtype=CALIB)
        n = off.size - 1
        means = np.empty(int(off[-1] - off[0]) + 16, dtype=np.float32)
        moff = np.empty(n + 1, dtype=np.uint64)
        info = np.zeros(n, dtype=EVT_INFO)
        _check(self.L, self.L.unc_detect_events(self.h, n, raw.ctypes.data, off.ctypes.data, cal.ctypes.data,
                                                means.ctypes.data, means.size, moff.ctypes.data, info.ctypes.data))
        return means[:int(moff[-1])], moff, info

    def trace(self, raw_i16, calib1, events_per_step=1, max_clusters=1 << 15):
        """Generator over Mapper::map_next steps of one read: (done, paths, clusters, max_map, len_sum, n_lens)."""
        raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
        cal = np.ascontiguousarray(calib1, dtype=CALIB)
        _check(self.L, self.L.unc_trace_begin(self.h, raw.ctypes.data, raw.size, cal.ctypes.data))
        mp = self.params.max_paths
        paths = np.zeros(mp, dtype=PATH)
        clus = np.zeros(max_clusters, dtype=CLUSTER)
        mm = np.zeros(1, dtype=CLUSTER)
        while True:
            done = C.c_int()
            _check(self.L, self.L.unc_trace_step(self.h, events_per_step, C.byref(done)))
            n, nc, nl, ls = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_float()
            _check(self.L, self.L.unc_trace_paths(self.h, paths.ctypes.data, mp, C.byref(n)))
            _check(self.L, self.L.unc_trace_clusters(self.h, clus.ctypes.data, max_clusters, C.byref(nc), mm.ctypes.data,
                                                     C.byref(ls), C.byref(nl)))
            yield bool(done.value), paths[:n.value].copy(), clus[:nc.value].copy(), mm[0].copy(), ls.value, nl.value
            if done.value:
                break

    def trace_finish(self):
        hit = np.zeros(1, dtype=HIT)
        _check(self.L, self.L.unc_trace_finish(self.h, hit.ctypes.data))
        return hit[0]


RT_TAP = np.dtype([("det_t", "<u4"), ("det_total_events", "<u4"), ("det_len_sum", "<f4"), ("norm_n", "<u4"), ("norm_wr", "<u4"),
                   ("prof_n", "<u4"), ("prof_to_mask", "<u4"), ("prof_queued", "<u4"), ("norm_mean", "<f8"), ("norm_varsum", "<f8"),
                   ("prof_mean", "<f8"), ("prof_varsum", "<f8"), ("prof_queue", "<f4", (28,))], align=True)


class Realtime:
    """Chunked path: RealtimePool + one Mapper per channel with MapPoolOrd's deterministic semantics
    (realtime_pool.cpp:74-142,349-358; map_pool_ord.cpp:61-112)."""

    def __init__(self, index, n_channels=512, params=None):
        self.index = index
        self.L = index.L
        self.params = params or default_params(self.L)
        self.n_channels = n_channels
        h = C.c_void_p()
        _check(self.L, self.L.unc_rt_create(index.h, C.byref(self.params), n_channels, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.unc_rt_free(self.h)
            self.h = None

    __del__ = close

    def device_bytes(self):
        return self.L.unc_rt_device_bytes(self.h)

    def tap_channel(self, channel):
        """unc_rt_tap_channel: (tap record, ring of 6000 floats) of one channel (parity tests)"""
        tap = np.zeros(1, dtype=RT_TAP)
        ring = np.zeros(6000, dtype=np.float32)
        _check(self.L, self.L.unc_rt_tap_channel(self.h, int(channel), tap.ctypes.data, ring.ctypes.data))
        return tap[0], ring

    def process_chunks(self, chunks, raw_i16=None, raw_ptr=None, stream=None, allow_overflow=False):
        """chunks: RT_CHUNK array (at most one per channel); raw: host int16 array or a device address."""
        ch = np.ascontiguousarray(chunks, dtype=RT_CHUNK)
        res = np.zeros(ch.size, dtype=RT_RESULT)
        if raw_ptr is not None:
            ptr, on_dev = C.c_void_p(raw_ptr), 1
        else:
            raw = np.ascontiguousarray(raw_i16, dtype=np.int16)
            ptr, on_dev = C.c_void_p(raw.ctypes.data), 0
        _check(self.L, self.L.unc_rt_process_chunks(self.h, ch.size, ch.ctypes.data, ptr, on_dev, C.c_void_p(stream or 0),
                                                    res.ctypes.data), allow=(UNC_ERR_OVERFLOW,) if allow_overflow else ())
        return res

    def process_chunks_f32(self, chunks, signal_f32):
        """the same for chunks that hold floats already (what the reference's Chunk keeps): `offset` indexes signal_f32"""
        ch = np.ascontiguousarray(chunks, dtype=RT_CHUNK)
        res = np.zeros(ch.size, dtype=RT_RESULT)
        sig = np.ascontiguousarray(signal_f32, dtype=np.float32)
        _check(self.L, self.L.unc_rt_process_chunks_f32(self.h, ch.size, ch.ctypes.data, C.c_void_p(sig.ctypes.data), 0, None,
                                                        res.ctypes.data))
        return res

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        self.L.unc_rt_last_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value
