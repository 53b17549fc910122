"""ctypes binding of ``libfqdedup_hip.so`` (C ABI: include/fqdedup_hip.h).

There is no CPU fallback anywhere in this package: if the library is missing or
no gfx950 device is visible, ``load()`` / ``Context()`` raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfqdedup_hip.so")

HOST, DEVICE, DEVICE_BORROW = 0, 1, 2
METRIC_HAMMING, METRIC_EDIT = 0, 1
METHODS = {"highest_count": 0, "adjacency": 1, "directional": 2}
T_PACK, T_COLLAPSE, T_EDGES, T_COMPONENTS, T_DISSECT, T_PAIRS_KERNEL, T_COUNT = 0, 1, 2, 3, 4, 5, 8

E_NOMEM, E_VALUE, E_LOOKUP, E_RUNTIME, E_DEVICE, E_STATE = -1, -2, -3, -4, -5, -6


class Summary(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_counted", C.c_uint64), ("n_unique", C.c_uint64),
                ("n_edges", C.c_uint64), ("n_clusters", C.c_uint64), ("n_kept", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Shape(C.Structure):
    _fields_ = [("planes", C.c_uint32), ("words", C.c_uint32), ("stride_words", C.c_uint32),
                ("max_len", C.c_uint32), ("ragged", C.c_uint32), ("alphabet_size", C.c_uint32),
                ("alphabet", C.c_uint8 * 128)]


# every symbol include/fqdedup_hip.h declares (tests check the library exports them all)
EXPORTS = [
    "fqd_device_count", "fqd_global_error", "fqd_create", "fqd_destroy", "fqd_last_error",
    "fqd_synchronize", "fqd_pack_keys", "fqd_configure", "fqd_scan_keys", "fqd_get_shape",
    "fqd_collapse", "fqd_find_edges", "fqd_components", "fqd_dissect", "fqd_cluster",
    "fqd_set_id_window", "fqd_get_kept_count", "fqd_get_kept_read_ids", "fqd_get_unique_table", "fqd_export_packed", "fqd_export_packed_by_owner", "fqd_import_packed",
    "fqd_export_packed_by_segment", "fqd_export_unique_by_segment", "fqd_gather_unique",
    "fqd_find_edges_segments", "fqd_edge_labels", "fqd_list_kept_except", "fqd_set_owner_rule", "fqd_declare_distinct_keys", "fqd_collapse_received", "fqd_set_kept_output",
    "fqd_export_unique", "fqd_import_unique", "fqd_export_edges", "fqd_import_edges",
    "fqd_within_distance", "fqd_contains", "fqd_quality_filter", "fqd_stage_times", "fqd_kernel_times", "fqd_set_timing", "fqd_cluster_keys", "fqd_edge_stats", "fqd_synth_keys",
    "fqd_store_add_keys", "fqd_store_remove", "fqd_store_removed_count", "fqd_get_clusters", "fqd_read_clusters",
    "fqd_trie_order", "fqd_trie_stats", "fqd_store_symbol_events", "fqd_get_stream", "fqd_pack_collapse", "fqd_synth_indel_keys", "fqd_copy_bandwidth",
    "fqd_synth_keys_skewed", "fqd_get_route", "fqd_cluster_subgraph", "fqd_cluster_subgraph_home", "fqd_dissect_except",
    "fqd_owner_routing_possible", "fqd_set_owner_routing", "fqd_dense_owner_slabs", "fqd_owner_slab_geometry", "fqd_pack_to_owner_slabs", "fqd_collapse_owner_slabs",
]

_lib: Optional[C.CDLL] = None


_pinned_live = [0]      # bytes of page-locked id arrays callers still hold


def _host_ids(n: int) -> np.ndarray:
    """A host array for n read ids. Large lists land in PAGE-LOCKED memory out of torch's caching host allocator (the
    array keeps its tensor alive; a released block is reused by the next call): the device-to-host copy then runs at
    the link's rate -- into a fresh pageable array 100 MB of ids took 9.8 ms (page faults + the driver's staging), into
    a pinned one 1.8 ms (tools/diag_e2e.py). A caller that KEEPS many results keeps that much memory locked: beyond
    FQD_PINNED_IDS_CAP bytes of live pinned results (default 4 GiB) further ones are pageable. FQD_NO_PINNED_IDS=1:
    always pageable."""
    if (1 << 17) <= n <= (1 << 27) and not os.environ.get("FQD_NO_PINNED_IDS"):
        cap = int(os.environ.get("FQD_PINNED_IDS_CAP", 4 << 30))
        if _pinned_live[0] + 8 * n <= cap:
            try:
                import torch
                import weakref
                if torch.cuda.is_available():
                    arr = torch.empty(n, dtype=torch.int64, pin_memory=True).numpy()
                    # (the array's base is the tensor object that keeps the block: it dies with the last view of the array)
                    keeper = arr.base if arr.base is not None else arr
                    _pinned_live[0] += 8 * n
                    weakref.finalize(keeper, lambda b=8 * n: _pinned_live.__setitem__(0, _pinned_live[0] - b))
                    return arr.view(np.uint64)
            except Exception:       # (no torch, or no page-locked memory to be had)
                pass
    return np.empty(n, dtype=np.uint64)


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m fastqdedup_amd.build` "
            "(hipcc --offload-arch=gfx950). fastqdedup_amd has no CPU fallback.")
    # torch bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's). Two HIP
    # runtimes in one process fight over the device, so when torch is installed it
    # is imported FIRST and the dynamic loader then binds this library to the
    # runtime torch already loaded. Without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # FQD_LIB_VARIANT=name: fastqdedup_amd/libfqdedup_hip.name.so instead (tools/ab_variants.sh: the same library built with
    # other compile-time switches, timed side by side on ONE box); a variant that is missing is an error, never a fallback
    path = LIB_PATH
    if os.environ.get("FQD_LIB_VARIANT"):
        path = LIB_PATH[:-3] + "." + os.environ["FQD_LIB_VARIANT"] + ".so"
        if not os.path.exists(path):
            raise ImportError(f"{path} is missing (FQD_LIB_VARIANT)")
    L = C.CDLL(path)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    L.fqd_device_count.restype = C.c_int
    L.fqd_global_error.restype = C.c_char_p
    L.fqd_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.fqd_destroy.argtypes = [vp]
    L.fqd_destroy.restype = None
    L.fqd_last_error.argtypes = [vp]
    L.fqd_last_error.restype = C.c_char_p
    L.fqd_synchronize.argtypes = [vp]
    L.fqd_pack_keys.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int]
    L.fqd_configure.argtypes = [vp, vp, C.c_uint32, C.c_int]
    L.fqd_scan_keys.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp,
                                C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.fqd_get_shape.argtypes = [vp, C.POINTER(Shape)]
    L.fqd_collapse.argtypes = [vp, vp, vp, C.c_int, u64p]
    L.fqd_find_edges.argtypes = [vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, u64p]
    L.fqd_components.argtypes = [vp, u64p]
    L.fqd_dissect.argtypes = [vp, C.c_int, u64p]
    L.fqd_cluster.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Summary)]
    L.fqd_cluster_keys.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.POINTER(Summary)]
    L.fqd_get_kept_read_ids.argtypes = [vp, vp, C.c_int]
    L.fqd_set_id_window.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.fqd_get_kept_count.argtypes = [vp, u64p, u64p]
    L.fqd_get_unique_table.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.fqd_export_packed.argtypes = [vp, vp, vp, vp, C.c_int]
    L.fqd_import_packed.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
    L.fqd_export_packed_by_owner.argtypes = [vp, C.c_uint32, C.c_uint64, vp, vp, vp, vp, vp, vp, C.c_int]
    L.fqd_export_packed_by_segment.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, vp, vp, vp, vp,
                                               vp, vp, C.c_int]
    L.fqd_export_unique_by_segment.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp,
                                               C.c_int]
    L.fqd_set_owner_rule.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L.fqd_declare_distinct_keys.argtypes = [vp]
    L.fqd_set_kept_output.argtypes = [vp, vp, C.c_uint64]
    L.fqd_collapse_received.argtypes = [vp, vp, u64p, u64p, C.c_uint32, C.c_uint64, C.c_int, u64p]
    L.fqd_gather_unique.argtypes = [vp, vp, C.c_uint64, vp, vp, vp, C.c_int]
    L.fqd_find_edges_segments.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, u64p]
    L.fqd_edge_labels.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, u64p, C.c_int]
    L.fqd_cluster_subgraph.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, vp, vp, u64p, u64p,
                                       C.c_int]
    L.fqd_cluster_subgraph_home.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, u64p, vp, vp, vp,
                                            u64p, u64p, u64p, u64p, C.c_int]
    L.fqd_dissect_except.argtypes = [vp, C.c_int, vp, C.c_uint64, C.c_int, u64p]
    L.fqd_dense_owner_slabs.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp]
    L.fqd_owner_routing_possible.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int)]
    L.fqd_set_owner_routing.argtypes = [vp, C.c_int]
    L.fqd_list_kept_except.argtypes = [vp, vp, C.c_uint64, C.c_int, u64p]
    L.fqd_export_unique.argtypes = [vp, vp, vp, vp, vp, C.c_int]
    L.fqd_import_unique.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_int]
    L.fqd_export_edges.argtypes = [vp, vp, C.c_int]
    L.fqd_import_edges.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.fqd_within_distance.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, C.c_int]
    L.fqd_contains.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, C.c_int, vp, C.c_int]
    L.fqd_quality_filter.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_double, vp, vp, vp,
                                     u64p, C.c_int]
    L.fqd_stage_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    L.fqd_kernel_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_int]
    L.fqd_set_timing.argtypes = [vp, C.c_int, C.c_uint32]
    L.fqd_edge_stats.argtypes = [vp, u64p, u64p, u64p]
    L.fqd_synth_keys.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                 C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]
    L.fqd_get_route.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.fqd_synth_keys_skewed.argtypes = [vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                        C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
    u32p = C.POINTER(C.c_uint32)
    L.fqd_owner_slab_geometry.argtypes = [C.c_uint64, C.c_uint32, u32p, u32p, u32p]
    L.fqd_pack_to_owner_slabs.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, u64p, C.POINTER(C.c_int)]
    L.fqd_collapse_owner_slabs.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u64p,
                                           C.c_uint64, C.c_uint64, C.c_uint32, u64p, C.POINTER(C.c_int)]
    L.fqd_synth_indel_keys.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                       C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp, vp]
    L.fqd_copy_bandwidth.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    L.fqd_get_stream.argtypes = [vp]
    L.fqd_get_stream.restype = C.c_void_p
    L.fqd_pack_collapse.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, C.c_int, C.c_uint32, u64p]
    L.fqd_store_add_keys.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint32, C.c_int, vp, vp, C.c_int, u64p]
    L.fqd_store_remove.argtypes = [vp, vp, C.c_uint64, C.c_int]
    L.fqd_store_removed_count.argtypes = [vp, u64p]
    L.fqd_get_clusters.argtypes = [vp, vp, C.c_uint32, u64p, u64p]
    L.fqd_read_clusters.argtypes = [vp, vp, vp, C.c_int]
    L.fqd_trie_order.argtypes = [vp, vp, C.c_uint32, vp, C.c_int]
    L.fqd_trie_stats.argtypes = [vp, vp, C.c_uint32, C.c_uint32, u64p, vp]
    L.fqd_store_symbol_events.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, C.c_uint32, vp, vp, vp, vp]
    _lib = L
    return L


def _raise(code: int, msg: str):
    if code == E_NOMEM:
        raise MemoryError(msg)
    if code == E_VALUE:
        raise ValueError(msg)
    if code == E_LOOKUP:
        raise LookupError(msg)
    raise RuntimeError(msg)


def _contiguous(x) -> bool:
    """True when _ptr_mem(x) points into x itself (not into a temporary contiguous copy)."""
    if x is None:
        return True
    if isinstance(x, np.ndarray):
        return bool(x.flags["C_CONTIGUOUS"])
    return bool(x.is_contiguous())


def _ptr_mem(x):
    """(pointer, mem kind, keepalive) of a numpy array (host) or torch tensor (device/host)."""
    if x is None:
        return None, HOST, None
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            x = np.ascontiguousarray(x)
        return x.ctypes.data if x.size else _EMPTY.ctypes.data, HOST, x
    if hasattr(x, "data_ptr"):  # torch tensor
        if not x.is_contiguous():
            x = x.contiguous()
        return x.data_ptr() or _EMPTY.ctypes.data, (DEVICE if x.is_cuda else HOST), x
    raise TypeError(f"expected numpy array or torch tensor, got {type(x).__name__}")


_EMPTY = np.zeros(16, dtype=np.uint8)


class Context:
    """One device + one HIP stream + the workspace of one clustering job."""

    def __init__(self, device: int = 0):
        L = load()
        h = C.c_void_p()
        rc = L.fqd_create(int(device), C.byref(h))
        if rc:
            _raise(rc, L.fqd_global_error().decode())
        self._h = h
        self._L = L
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.fqd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc: int):
        if rc:
            _raise(rc, self._L.fqd_last_error(self._h).decode())

    # ---- stages ---------------------------------------------------------------
    def _same_mem(self, *kinds):
        ks = {k for k, present in kinds if present}
        if len(ks) > 1:
            raise ValueError("all buffers of one call must live on the same side (host or device)")
        return ks.pop() if ks else HOST

    def pack_keys(self, keys, offsets=None, key_len: int = 0):
        kp, km, _k = _ptr_mem(keys)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
            if key_len <= 0:
                if nbytes:
                    raise ValueError("key_len must be positive when offsets is None")
                n = 0
            else:
                if nbytes % key_len:
                    raise ValueError("key buffer is not a multiple of key_len")
                n = nbytes // key_len
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
            if n < 0:
                raise ValueError("offsets needs n+1 entries")
        mem = self._same_mem((km, True), (om, offsets is not None))
        self._ck(self._L.fqd_pack_keys(self._h, kp, op, n, int(key_len), mem))
        return n

    @staticmethod
    def owner_slab_geometry(n_max: int, n_parts: int):
        """(hash bins per owner, slabs per bin, records per slab) of the fused multi-GPU way in."""
        hb, subs, cap = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        rc = load().fqd_owner_slab_geometry(int(n_max), int(n_parts), C.byref(hb), C.byref(subs), C.byref(cap))
        if rc:
            raise ValueError("bad owner slab geometry")
        return int(hb.value), int(subs.value), int(cap.value)

    def owner_routing_possible(self, key_len: int, n_segments: int) -> bool:
        v = C.c_int(0)
        self._ck(self._L.fqd_owner_routing_possible(self._h, int(key_len), int(n_segments), C.byref(v)))
        return bool(v.value)

    def set_owner_routing(self, enable: bool):
        self._ck(self._L.fqd_set_owner_routing(self._h, 1 if enable else 0))

    def pack_to_owner_slabs(self, keys, key_len: int, n_parts: int, n_segments: int, segment: int, geometry,
                            slabs_out, cursors_out):
        """fqd_pack_to_owner_slabs -> reads per owner (list), or None when the general way must be taken."""
        kp, km, _k = _ptr_mem(keys)
        nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
        n = nbytes // key_len if key_len > 0 else 0
        sp, _sm, _s = _ptr_mem(slabs_out)
        cp, _cm, _c = _ptr_mem(cursors_out)
        hb, subs, cap = geometry
        counts = (C.c_uint64 * n_parts)()
        done = C.c_int(0)
        self._ck(self._L.fqd_pack_to_owner_slabs(self._h, kp, n, int(key_len), km, int(n_parts), int(n_segments),
                                                 int(segment), hb, subs, cap, sp, cp, counts, C.byref(done)))
        return [int(x) for x in counts] if done.value else None

    def dense_owner_slabs(self, slabs, cursors, n_parts: int, geometry, rows_out, fills_out):
        """fqd_dense_owner_slabs: the filled prefixes of the slabs back to back + every slab's fill (device buffers)."""
        sp, _m, _0 = _ptr_mem(slabs)
        cp, _m, _1 = _ptr_mem(cursors)
        rp, _m, _2 = _ptr_mem(rows_out)
        fp, _m, _3 = _ptr_mem(fills_out)
        hb, subs, cap = geometry
        room = int(rows_out.shape[0]) if hasattr(rows_out, "shape") and len(rows_out.shape) == 2 else int(rows_out.numel()) // 4
        self._ck(self._L.fqd_dense_owner_slabs(self._h, sp, cp, int(n_parts), hb, subs, cap, rp, room, fp))

    def collapse_owner_slabs(self, slabs, cursors, n_senders: int, my_part: int, geometry, sender_id0, id_limit: int,
                             n_reads: int, search_segments: int = 0):
        """fqd_collapse_owner_slabs -> unique keys, or None when the general way must be taken."""
        sp, _sm, _s = _ptr_mem(slabs)
        cp, _cm, _c = _ptr_mem(cursors)
        hb, subs, cap = geometry
        ids = (C.c_uint64 * n_senders)(*[int(x) for x in sender_id0])
        u, done = C.c_uint64(0), C.c_int(0)
        self._ck(self._L.fqd_collapse_owner_slabs(self._h, sp, cp, int(n_senders), int(my_part), hb, subs, cap, ids,
                                                  int(id_limit), int(n_reads), int(search_segments), C.byref(u),
                                                  C.byref(done)))
        return int(u.value) if done.value else None

    def stream_handle(self) -> int:
        """The context's hipStream_t as an integer (torch.cuda.ExternalStream takes it)."""
        return int(self._L.fqd_get_stream(self._h) or 0)

    def pack_collapse(self, keys, offsets=None, key_len: int = 0, weights=None, read_ids=None,
                      search_segments: int = 0) -> int:
        """pack_keys + collapse in one C call (fqd_pack_collapse) -> unique keys."""
        kp, km, _k = _ptr_mem(keys)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
            n = nbytes // key_len if key_len > 0 else 0
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
        mem = self._same_mem((km, True), (om, offsets is not None))
        wp, wm, _w = _ptr_mem(weights)
        rp, rm, _r = _ptr_mem(read_ids)
        aux = self._same_mem((wm, weights is not None), (rm, read_ids is not None))
        u = C.c_uint64(0)
        self._ck(self._L.fqd_pack_collapse(self._h, kp, op, n, int(key_len), mem, wp, rp, aux, int(search_segments),
                                           C.byref(u)))
        return int(u.value)

    def configure(self, present128: Optional[np.ndarray], max_len: int = 0, ragged: bool = False):
        if present128 is None:
            self._ck(self._L.fqd_configure(self._h, None, 0, 0))
            return
        p = np.ascontiguousarray(present128, dtype=np.uint8)
        assert p.size == 128
        self._ck(self._L.fqd_configure(self._h, p.ctypes.data, int(max_len), int(bool(ragged))))

    def scan_keys(self, keys, offsets=None, key_len: int = 0):
        kp, km, _k = _ptr_mem(keys)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
            n = nbytes // key_len if key_len else 0
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
        mem = self._same_mem((km, True), (om, offsets is not None))
        present = np.zeros(128, dtype=np.uint8)
        max_len, ragged = C.c_uint32(0), C.c_int(0)
        self._ck(self._L.fqd_scan_keys(self._h, kp, op, n, int(key_len), mem, present.ctypes.data,
                                       C.byref(max_len), C.byref(ragged)))
        return present, int(max_len.value), bool(ragged.value)

    def shape(self) -> Shape:
        s = Shape()
        self._ck(self._L.fqd_get_shape(self._h, C.byref(s)))
        return s

    def collapse(self, weights=None, read_ids=None) -> int:
        wp, wm, _w = _ptr_mem(weights)
        rp, rm, _r = _ptr_mem(read_ids)
        mem = self._same_mem((wm, weights is not None), (rm, read_ids is not None))
        u = C.c_uint64(0)
        self._ck(self._L.fqd_collapse(self._h, wp, rp, mem, C.byref(u)))
        return int(u.value)

    def find_edges(self, max_distance: int, metric: int = METRIC_HAMMING, shard: int = 0,
                   n_shards: int = 1) -> int:
        e = C.c_uint64(0)
        self._ck(self._L.fqd_find_edges(self._h, int(max_distance), int(metric), int(shard),
                                        int(n_shards), C.byref(e)))
        return int(e.value)

    def components(self) -> int:
        v = C.c_uint64(0)
        self._ck(self._L.fqd_components(self._h, C.byref(v)))
        return int(v.value)

    def dissect(self, method: int) -> int:
        v = C.c_uint64(0)
        self._ck(self._L.fqd_dissect(self._h, int(method), C.byref(v)))
        return int(v.value)

    def cluster(self, weights=None, read_ids=None, *, max_distance: int = 1,
                metric: int = METRIC_HAMMING, method: int = 2) -> dict:
        wp, wm, _w = _ptr_mem(weights)
        rp, rm, _r = _ptr_mem(read_ids)
        mem = self._same_mem((wm, weights is not None), (rm, read_ids is not None))
        s = Summary()
        self._ck(self._L.fqd_cluster(self._h, wp, rp, mem, int(max_distance), int(metric),
                                     int(method), C.byref(s)))
        return s.as_dict()

    def cluster_keys(self, keys, offsets=None, key_len: int = 0, weights=None, read_ids=None, *,
                     max_distance: int = 1, metric: int = METRIC_HAMMING, method: int = 2) -> dict:
        """pack_keys + cluster in one C call (fqd_cluster_keys): the pack kernel may then feed the
        collapse directly (short fixed-length keys)."""
        kp, km, _k = _ptr_mem(keys)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
            if key_len <= 0:
                if nbytes:
                    raise ValueError("key_len must be positive when offsets is None")
                n = 0
            else:
                if nbytes % key_len:
                    raise ValueError("key buffer is not a multiple of key_len")
                n = nbytes // key_len
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
            if n < 0:
                raise ValueError("offsets needs n+1 entries")
        mem = self._same_mem((km, True), (om, offsets is not None))
        wp, wm, _w = _ptr_mem(weights)
        rp, rm, _r = _ptr_mem(read_ids)
        aux = self._same_mem((wm, weights is not None), (rm, read_ids is not None))
        s = Summary()
        self._ck(self._L.fqd_cluster_keys(self._h, kp, op, n, int(key_len), mem, wp, rp, aux, int(max_distance),
                                          int(metric), int(method), C.byref(s)))
        return s.as_dict()

    # ---- results --------------------------------------------------------------
    def set_id_window(self, lo: int = 0, hi: int = 0xFFFFFFFFFFFFFFFF):
        self._ck(self._L.fqd_set_id_window(self._h, int(lo), int(hi)))

    def kept_count(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.fqd_get_kept_count(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def kept_read_ids(self, n_kept: int, out=None):
        if out is None:
            out = _host_ids(n_kept)
        op, om, _o = _ptr_mem(out)
        self._ck(self._L.fqd_get_kept_read_ids(self._h, op, om))
        return out

    def unique_table(self, n_unique: int, labels: bool = True, kept: bool = True):
        first = np.empty(n_unique, dtype=np.uint64)
        counts = np.empty(n_unique, dtype=np.uint32)
        lab = np.empty(n_unique, dtype=np.uint32) if labels else None
        kp = np.empty(n_unique, dtype=np.uint8) if kept else None
        self._ck(self._L.fqd_get_unique_table(
            self._h, first.ctypes.data, counts.ctypes.data,
            lab.ctypes.data if labels else None, kp.ctypes.data if kept else None, HOST))
        return first, counts, lab, kp

    def labels_into(self, labels):
        """The component label (smallest row of its component) of every unique key (uint32) into a caller's buffer."""
        lp, lm, _0 = _ptr_mem(labels)
        self._ck(self._L.fqd_get_unique_table(self._h, None, None, lp, None, lm))

    def kept_flags_into(self, kept):
        """The dissection's verdict per unique key (uint8) into a caller's buffer."""
        kp, km, _0 = _ptr_mem(kept)
        self._ck(self._L.fqd_get_unique_table(self._h, None, None, None, kp, km))

    # ---- exchange (raw pointers; the caller owns the buffers) ----------------------
    def export_packed(self, recs, lens, hashes):
        rp, rm, _1 = _ptr_mem(recs)
        lp, lm, _2 = _ptr_mem(lens)
        hp, hm, _3 = _ptr_mem(hashes)
        self._ck(self._L.fqd_export_packed(self._h, rp, lp, hp, rm))

    def export_packed_by_owner(self, n_parts: int, id0: int, weights, recs, lens, ids, weights_out):
        wp, _m, _0 = _ptr_mem(weights)
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        ip, _m, _3 = _ptr_mem(ids)
        op, _m, _4 = _ptr_mem(weights_out)
        counts = np.zeros(n_parts, dtype=np.uint64)
        self._ck(self._L.fqd_export_packed_by_owner(self._h, int(n_parts), int(id0), wp, rp, lp, ip, op,
                                                    counts.ctypes.data, rm))
        return counts

    def export_packed_by_segment(self, n_parts: int, n_segments: int, segment: int, id0: int, weights, recs, lens,
                                 ids, weights_out):
        wp, _m, _0 = _ptr_mem(weights)
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        ip, _m, _3 = _ptr_mem(ids)
        op, _m, _4 = _ptr_mem(weights_out)
        counts = np.zeros(n_parts, dtype=np.uint64)
        self._ck(self._L.fqd_export_packed_by_segment(self._h, int(n_parts), int(n_segments), int(segment), int(id0),
                                                      wp, rp, lp, ip, op, counts.ctypes.data, rm))
        return counts

    def set_owner_rule(self, n_parts: int, n_segments: int = 1, segment: int = 0):
        """pack_keys also computes every read's owner rank (0 parts: off)."""
        self._ck(self._L.fqd_set_owner_rule(self._h, int(n_parts), int(n_segments), int(segment)))

    def export_unique_by_segment(self, n_parts: int, n_segments: int, segment: int, uid_base: int, recs, lens, uids):
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        up, _m, _3 = _ptr_mem(uids)
        counts = np.zeros(n_parts, dtype=np.uint64)
        self._ck(self._L.fqd_export_unique_by_segment(self._h, int(n_parts), int(n_segments), int(segment),
                                                      int(uid_base), rp, lp, up, counts.ctypes.data, rm))
        return counts

    def gather_unique(self, idx, n: int, recs, lens, counts):
        ip, im, _0 = _ptr_mem(idx)
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        cp, _m, _3 = _ptr_mem(counts)
        self._ck(self._L.fqd_gather_unique(self._h, ip, int(n), rp, lp, cp, rm))

    def find_edges_segments(self, max_distance: int, seg_lo: int, seg_hi: int) -> int:
        ne = C.c_uint64(0)
        self._ck(self._L.fqd_find_edges_segments(self._h, int(max_distance), int(seg_lo), int(seg_hi), C.byref(ne)))
        return ne.value

    def edge_labels(self, uv, n_edges: int, n_nodes: int, roots) -> int:
        """roots[e] = smallest node of edge e's component; returns the number of components."""
        ep, em, _0 = _ptr_mem(uv)
        rp, rm, _1 = _ptr_mem(roots)
        nc = C.c_uint64(0)
        self._ck(self._L.fqd_edge_labels(self._h, ep, int(n_edges), int(n_nodes), rp, C.byref(nc), DEVICE))
        return nc.value

    def cluster_subgraph(self, uv, roots, n_edges: int, n_nodes: int, n_parts: int, part: int, touched_out, sub_out):
        """fqd_cluster_subgraph -> (number of touched nodes, number of edges of this part)."""
        ep, _m, _0 = _ptr_mem(uv)
        rp, _m, _1 = _ptr_mem(roots)
        tp, _m, _2 = _ptr_mem(touched_out)
        sp, _m, _3 = _ptr_mem(sub_out)
        nt, ns = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.fqd_cluster_subgraph(self._h, ep, rp, int(n_edges), int(n_nodes), int(n_parts), int(part), tp,
                                              sp, C.byref(nt), C.byref(ns), DEVICE))
        return int(nt.value), int(ns.value)

    def cluster_subgraph_home(self, uv, roots, n_edges: int, n_nodes: int, n_parts: int, part: int, uid_bounds,
                              touched_out, sub_out, home_out):
        """fqd_cluster_subgraph_home -> (touched nodes, edges of this part's spanning clusters, home edges, edges of
        ALL parts' spanning clusters)."""
        ep, _m, _0 = _ptr_mem(uv)
        rp, _m, _1 = _ptr_mem(roots)
        tp, _m, _2 = _ptr_mem(touched_out)
        sp, _m, _3 = _ptr_mem(sub_out)
        hp, _m, _4 = _ptr_mem(home_out)
        bounds = (C.c_uint64 * (n_parts + 1))(*[int(b) for b in uid_bounds])
        nt, ns, nh, nsp = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.fqd_cluster_subgraph_home(self._h, ep, rp, int(n_edges), int(n_nodes), int(n_parts), int(part),
                                                   bounds, tp, sp, hp, C.byref(nt), C.byref(ns), C.byref(nh),
                                                   C.byref(nsp), DEVICE))
        return int(nt.value), int(ns.value), int(nh.value), int(nsp.value)

    def dissect_except(self, method: int, dropped, n_dropped: int) -> int:
        dp, _m, _0 = _ptr_mem(dropped)
        nk = C.c_uint64(0)
        self._ck(self._L.fqd_dissect_except(self._h, int(method), dp, int(n_dropped), DEVICE, C.byref(nk)))
        return int(nk.value)

    def list_kept_except(self, dropped, n_dropped: int) -> int:
        dp, dm, _0 = _ptr_mem(dropped)
        nk = C.c_uint64(0)
        self._ck(self._L.fqd_list_kept_except(self._h, dp, int(n_dropped), DEVICE, C.byref(nk)))
        return nk.value

    def import_packed(self, recs, lens, n: int, borrow: bool = False):
        """borrow: read the (device) buffers in place; keep them alive until collapse() returned."""
        borrow = borrow and _contiguous(recs) and _contiguous(lens)
        rp, rm, _1 = _ptr_mem(recs)
        lp, lm, _2 = _ptr_mem(lens)
        self._ck(self._L.fqd_import_packed(self._h, rp, lp, int(n), DEVICE_BORROW if borrow and rm == DEVICE else rm))

    def export_unique(self, recs, lens, counts, first_ids):
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        cp, _m, _3 = _ptr_mem(counts)
        fp, _m, _4 = _ptr_mem(first_ids)
        self._ck(self._L.fqd_export_unique(self._h, rp, lp, cp, fp, rm))

    def import_unique(self, recs, lens, counts, first_ids, n_unique: int, borrow: bool = False):
        """borrow: records/lengths are read in place; keep them alive while this table is in use."""
        borrow = borrow and _contiguous(recs) and _contiguous(lens)
        rp, rm, _1 = _ptr_mem(recs)
        lp, _m, _2 = _ptr_mem(lens)
        cp, _m, _3 = _ptr_mem(counts)
        fp, _m, _4 = _ptr_mem(first_ids)
        self._ck(self._L.fqd_import_unique(self._h, rp, lp, cp, fp, int(n_unique),
                                           DEVICE_BORROW if borrow and rm == DEVICE else rm))

    def collapse_received(self, weights, seg_rows, seg_id0, id_limit: int = 2**64 - 1) -> int:
        """Collapse reads received from several ranks whose records carry the sender's local read
        index in their padding word (export_packed_by_segment with ids=None): rows
        seg_rows[s]..seg_rows[s+1] came from the rank whose first read has id seg_id0[s]."""
        wp, wm, _0 = _ptr_mem(weights)
        rows = np.ascontiguousarray(seg_rows, dtype=np.uint64)
        id0 = np.ascontiguousarray(seg_id0, dtype=np.uint64)
        nu = C.c_uint64(0)
        self._ck(self._L.fqd_collapse_received(self._h, wp, rows.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               id0.ctypes.data_as(C.POINTER(C.c_uint64)), len(id0),
                                               int(id_limit), wm if weights is not None else DEVICE, C.byref(nu)))
        return nu.value

    def set_kept_output(self, out=None):
        """Have the next dissection write its kept-id list straight into `out` (device int64/uint64
        tensor with room for any outcome); None switches it off."""
        if out is None:
            self._ck(self._L.fqd_set_kept_output(self._h, None, 0))
            return
        p, m, _k = _ptr_mem(out)
        if m != DEVICE:
            raise ValueError("kept output buffer must be a device tensor")
        self._ck(self._L.fqd_set_kept_output(self._h, p, int(out.numel())))

    def declare_distinct_keys(self):
        """The imported rows hold pairwise distinct keys (rows of other ranks' collapsed tables)."""
        self._ck(self._L.fqd_declare_distinct_keys(self._h))

    def export_edges(self, uv):
        p, m, _ = _ptr_mem(uv)
        self._ck(self._L.fqd_export_edges(self._h, p, m))

    def import_edges(self, uv, n_edges: int):
        p, m, _ = _ptr_mem(uv)
        self._ck(self._L.fqd_import_edges(self._h, p, int(n_edges), m))

    # ---- single calls ---------------------------------------------------------
    def within_distance(self, a_bytes, a_off, b_bytes, b_off, max_distance: int, metric: int):
        n = len(a_off) - 1
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self._ck(self._L.fqd_within_distance(
            self._h, a_bytes.ctypes.data, a_off.ctypes.data, b_bytes.ctypes.data, b_off.ctypes.data,
            n, int(max_distance), int(metric), out.ctypes.data, HOST))
        return out[:n]

    def contains(self, q_bytes, q_off, max_distance: int, metric: int):
        n = len(q_off) - 1
        out = np.zeros(max(n, 1), dtype=np.uint8)
        self._ck(self._L.fqd_contains(self._h, q_bytes.ctypes.data, q_off.ctypes.data, n,
                                      int(max_distance), int(metric), out.ctypes.data, HOST))
        return out[:n]

    # ---- quality gate -----------------------------------------------------------
    def quality_filter(self, quals, offsets=None, qual_len: int = 0, *, threshold: float = 0.001,
                       phred_offset: int = 33, want_means: bool = False, table=None):
        """pass flags (uint32, 1 = keep) [, means], number discarded."""
        qp, qm, _q = _ptr_mem(quals)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = quals.numel() if hasattr(quals, "numel") else quals.size
            n = nbytes // qual_len if qual_len else 0
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
        mem = self._same_mem((qm, True), (om, offsets is not None))
        if mem == DEVICE:
            import torch
            dev = quals.device
            flags = torch.empty(n, dtype=torch.int32, device=dev)
            means = torch.empty(n, dtype=torch.float64, device=dev) if want_means else None
        else:
            flags = np.empty(n, dtype=np.uint32)
            means = np.empty(n, dtype=np.float64) if want_means else None
        fp, _m, _f = _ptr_mem(flags)
        mp_, _m, _mm = _ptr_mem(means)
        tb = None if table is None else np.ascontiguousarray(table, dtype=np.float64)
        nd = C.c_uint64(0)
        self._ck(self._L.fqd_quality_filter(self._h, qp, op, n, int(qual_len), int(phred_offset),
                                            float(threshold), None if tb is None else tb.ctypes.data,
                                            fp, mp_, C.byref(nd), mem))
        return flags, means, int(nd.value)

    # ---- the Trie object as a device-resident store ------------------------------
    @staticmethod
    def _alphabet_bytes(alphabet: str) -> np.ndarray:
        return np.frombuffer(alphabet.encode("latin-1") or b"\0", dtype=np.uint8)

    def store_add_keys(self, keys, offsets=None, key_len: int = 0, weights=None, read_ids=None) -> int:
        """Merge new keys into the resident unique table (fqd_store_add_keys) -> unique keys stored."""
        kp, km, _k = _ptr_mem(keys)
        op, om, _o = _ptr_mem(offsets)
        if offsets is None:
            nbytes = keys.numel() if hasattr(keys, "numel") else keys.size
            n = nbytes // key_len if key_len > 0 else 0
        else:
            n = (offsets.numel() if hasattr(offsets, "numel") else offsets.size) - 1
        mem = self._same_mem((km, True), (om, offsets is not None))
        wp, wm, _w = _ptr_mem(weights)
        rp, rm, _r = _ptr_mem(read_ids)
        aux = self._same_mem((wm, weights is not None), (rm, read_ids is not None))
        u = C.c_uint64(0)
        self._ck(self._L.fqd_store_add_keys(self._h, kp, op, n, int(key_len), mem, wp, rp, aux, C.byref(u)))
        return int(u.value)

    def store_remove(self, uids) -> None:
        up, um, _u = _ptr_mem(uids)
        n = uids.numel() if hasattr(uids, "numel") else uids.size
        self._ck(self._L.fqd_store_remove(self._h, up, int(n), um))

    def store_removed_count(self) -> int:
        v = C.c_uint64(0)
        self._ck(self._L.fqd_store_removed_count(self._h, C.byref(v)))
        return int(v.value)

    def clusters(self, alphabet: str):
        """(offsets uint64[n_clusters + 1], member uids uint32[n_members]) in pop_cluster order."""
        ab = self._alphabet_bytes(alphabet)
        nc, nm = C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.fqd_get_clusters(self._h, ab.ctypes.data, len(alphabet), C.byref(nc), C.byref(nm)))
        offsets = np.zeros(int(nc.value) + 1, dtype=np.uint64)
        members = np.zeros(int(nm.value), dtype=np.uint32)
        self._ck(self._L.fqd_read_clusters(self._h, offsets.ctypes.data,
                                           members.ctypes.data if members.size else None, HOST))
        return offsets, members

    def store_symbol_events(self, alphabet: str, symbols: str, after, reuse_order: bool):
        """One round of the lazy-alphabet search (fqd_store_symbol_events): per symbol
        (candidate first id | None, depth, partner first id | None)."""
        ab, sy = self._alphabet_bytes(alphabet), self._alphabet_bytes(symbols)
        n = len(symbols)
        none = 0xFFFFFFFFFFFFFFFF
        aft = np.array([none if a is None else int(a) for a in after], dtype=np.uint64)
        cand = np.zeros(max(n, 1), dtype=np.uint64)
        depth = np.zeros(max(n, 1), dtype=np.uint32)
        partner = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self._L.fqd_store_symbol_events(self._h, ab.ctypes.data, len(alphabet), int(bool(reuse_order)),
                                                 sy.ctypes.data, n, aft.ctypes.data, cand.ctypes.data,
                                                 depth.ctypes.data, partner.ctypes.data))
        return [(None if int(cand[i]) == none else int(cand[i]), int(depth[i]),
                 None if int(partner[i]) == none else int(partner[i])) for i in range(n)]

    def trie_order(self, alphabet: str, n_unique: int) -> np.ndarray:
        ab = self._alphabet_bytes(alphabet)
        out = np.zeros(n_unique, dtype=np.uint32)
        self._ck(self._L.fqd_trie_order(self._h, ab.ctypes.data, len(alphabet),
                                        out.ctypes.data if n_unique else None, HOST))
        return out

    def trie_stats(self, alphabet: str, n_layers: int):
        """(memory_size, raw_stats as n_layers lists of len(alphabet) + 1 counters)."""
        ab = self._alphabet_bytes(alphabet)
        cols = len(alphabet) + 1
        stats = np.zeros(max(n_layers * cols, 1), dtype=np.uint64)
        mem = C.c_uint64(0)
        self._ck(self._L.fqd_trie_stats(self._h, ab.ctypes.data, len(alphabet), int(n_layers), C.byref(mem),
                                        stats.ctypes.data))
        rows = stats[: n_layers * cols].reshape(n_layers, cols)
        return int(mem.value), [[int(v) for v in row] for row in rows]

    # ---- measurement ------------------------------------------------------------
    def stage_times(self):
        ms = (C.c_float * T_COUNT)()
        ln = (C.c_uint32 * T_COUNT)()
        self._ck(self._L.fqd_stage_times(self._h, ms, ln))
        names = ["pack", "collapse", "edges", "components", "dissect"]   # per kernel: kernel_times()
        return ({k: float(ms[i]) for i, k in enumerate(names)},
                {k: int(ln[i]) for i, k in enumerate(names)})

    KERNELS = ["pack_kernel", "part_hist_kernel<1>", "part_scatter_kernel<1>", "part_hist_kernel<2>",
               "part_scatter_kernel<2>", "bucket_dedupe_kernel", "bucket_compact_kernel", "head_flags_kernel",
               "write_unique_kernel", "segment_hashes_kernel", "bucket_pairs_kernel", "uf_union_kernel",
               "uf_flatten_kernel", "dissect_round_kernel", "gp_hist_kernel", "gp_scatter_kernel", "verify_candidates_kernel", "kept_flags_kernel",
               "part_scatter12_kernel", "bucket_dedupe12_kernel"]

    def kernel_times(self, reset: bool = True):
        """{kernel: (ms summed over launches, launches)} since the last reset."""
        ms = (C.c_float * 20)()
        ln = (C.c_uint32 * 20)()
        self._ck(self._L.fqd_kernel_times(self._h, ms, ln, int(reset)))
        return {k: (float(ms[i]), int(ln[i])) for i, k in enumerate(self.KERNELS)}

    def set_timing(self, stage_timers: bool = True, kernels=None):
        """Which timers record events: the stage timers, and the kernels named in ``kernels``
        (None = all of KERNELS, () = none)."""
        mask = 0xFFFFFFFF if kernels is None else sum(1 << self.KERNELS.index(k) for k in kernels)
        self._ck(self._L.fqd_set_timing(self._h, int(bool(stage_timers)), mask))

    def edge_stats(self):
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._ck(self._L.fqd_edge_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"keys_gathered": int(a.value), "pairs_compared": int(b.value), "edges": int(c.value)}

    def copy_bandwidth(self, src, dst, reps: int = 5) -> float:
        """GB/s (read + write) of a device-to-device copy of ``src`` into ``dst`` (fqd_copy_bandwidth)."""
        nbytes = src.numel() * src.element_size()
        out = C.c_double(0.0)
        self._ck(self._L.fqd_copy_bandwidth(self._h, src.data_ptr(), dst.data_ptr(), int(nbytes) & ~15, int(reps),
                                            C.byref(out)))
        return float(out.value)

    def synth_indel_keys(self, n_total: int, start: int, count: int, length: int, umi: int, seed: int,
                         indel_rate: float = 0.01, copies: int = 4, sub_rate: float = 1e-3, n_rate: float = 1e-4):
        """(key bytes uint8, offsets int64[count + 1]) on this context's device: the fixed-length job
        of synth_keys with an indel tail (three key lengths), byte-identical to synth.indel_variant."""
        import torch
        from .synth import rate_threshold
        dev = torch.device("cuda", self.device)
        args = (n_total, start, count, length, umi, seed, copies, rate_threshold(n_rate), rate_threshold(sub_rate),
                rate_threshold(indel_rate))
        lens = torch.empty(max(count, 1), dtype=torch.int64, device=dev)
        self._ck(self._L.fqd_synth_indel_keys(self._h, *args, lens.data_ptr(), None, None))
        offsets = torch.zeros(count + 1, dtype=torch.int64, device=dev)
        if count:
            torch.cumsum(lens[:count], 0, out=offsets[1:])
        total = int(offsets[-1].item())
        out = torch.empty(max(total, 16), dtype=torch.uint8, device=dev)
        self._ck(self._L.fqd_synth_indel_keys(self._h, *args, None, offsets.data_ptr(), out.data_ptr()))
        return out[:total], offsets

    def synth_keys(self, out_tensor, n_total: int, start: int, count: int, length: int, umi: int,
                   seed: int, copies: int = 4, sub_rate: float = 1e-3, n_rate: float = 1e-4, skew=None):
        from .synth import rate_threshold
        if skew:
            self._ck(self._L.fqd_synth_keys_skewed(self._h, out_tensor.data_ptr(), n_total, start, count, length,
                                                   umi, seed, copies, rate_threshold(n_rate), rate_threshold(sub_rate),
                                                   rate_threshold(skew.get("hot", 0.0)),
                                                   rate_threshold(skew.get("ladder", 0.0)),
                                                   int(skew.get("lowc_every", 0))))
            return
        self._ck(self._L.fqd_synth_keys(self._h, out_tensor.data_ptr(), n_total, start, count, length,
                                        umi, seed, copies, rate_threshold(n_rate),
                                        rate_threshold(sub_rate)))

    ROUTE_BITS = {"fused_pack": 0x1, "compact_records": 0x2, "pass0_in_collapse": 0x4, "restarted": 0x8,
                  "collapse_lds": 0x10, "collapse_pairs": 0x20, "collapse_sort": 0x40, "search_grouped": 0x100,
                  "search_sort": 0x200, "search_edit": 0x400, "search_retried": 0x800, "pass0_continued": 0x1000,
                  "spill_list": 0x2000, "search_refined": 0x4000, "one_kernel_collapse": 0x8000, "search_tiles": 0x10000}

    def route(self) -> dict:
        """Which way the last job took (fqd_get_route): {name: bool} over the FQD_ROUTE_* bits."""
        v = C.c_uint32(0)
        self._ck(self._L.fqd_get_route(self._h, C.byref(v)))
        return {k: bool(v.value & b) for k, b in self.ROUTE_BITS.items()}

    def synchronize(self):
        self._ck(self._L.fqd_synchronize(self._h))
