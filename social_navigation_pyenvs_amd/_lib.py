"""ctypes binding of libcrowdstep.so (the C ABI declared in include/crowdstep.h).

There is no CPU fallback: if the HIP library is missing or no GPU is visible, every compute entry
point raises ``CrowdstepError``.  Loading the library (symbol table) works without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# CROWDSTEP_LIB: another build of the library (same-box A/B of two kernel builds: tools/ab_libs.sh, tools/ab_vs_prev.sh); never set in production
LIB_PATH = os.environ.get("CROWDSTEP_LIB") or os.path.join(_PKG, "libcrowdstep.so")

CS_OK = 0
CS_ERR_ARG, CS_ERR_TYPE, CS_ERR_HIP, CS_ERR_NO_DEVICE = -1, -2, -3, -4
CS_LAYOUT_AOS, CS_LAYOUT_SOA = 0, 1
CS_ALL_PARAMS_EQUAL = 1 << 0
CS_ROBOT_ROW = 1 << 1
CS_PARAMS_SHARED = 1 << 2
CS_OBSTACLES_SHARED = 1 << 3
CS_RESPAWN = 1 << 4
CS_ROBOT_UNICYCLE = 1 << 5
CS_ORCA = 9
CS_SOCIAL_MOMENTUM = 10

ABI_VERSION = 4   # include/crowdstep.h CS_ABI_VERSION
# cs_worlds.orca_math
CS_ORCA_MATH_DEFAULT, CS_ORCA_MATH_EXACT, CS_ORCA_MATH_FAST, CS_ORCA_MATH_FMA = 0, 1, 2, 3
ORCA_MATH_NAMES = {"default": 0, "exact": 1, "fast": 2, "fma": 3}

# every symbol include/crowdstep.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "cs_last_error", "cs_abi_version", "cs_device_count", "cs_set_device", "cs_device_name", "cs_device_pci_bus_id", "cs_malloc",
    "cs_free", "cs_memcpy_h2d", "cs_memcpy_d2h", "cs_memcpy_d2d", "cs_memset", "cs_stream_create", "cs_stream_create_with_priority",
    "cs_stream_destroy", "cs_stream_sync", "cs_event_create", "cs_event_destroy", "cs_event_record",
    "cs_event_elapsed_ms", "cs_event_query", "cs_stream_wait_event", "cs_graph_begin_capture", "cs_graph_end_capture", "cs_graph_launch", "cs_graph_destroy",
    "cs_update_humans_parallel", "cs_step", "cs_peek", "cs_collision_reward",
    "cs_state_aos_to_soa", "cs_state_soa_to_aos", "cs_launch_geometry", "cs_lookahead",
    "cs_generate_scratch_bytes", "cs_generate_worlds", "cs_laser_scan", "cs_robot_model_step", "cs_actual_collision_reward",
    "cs_update_humans_rk45", "cs_gym_bookkeeping", "cs_step_variant", "cs_debug_divsqrt_check", "cs_gym_observe", "cs_copy_worlds_masked", "cs_imitation_block", "cs_gym_bookkeeping_next_step", "cs_robot_model_velocities",
    "cs_step_trace", "cs_reserve_scratch", "cs_release_scratch", "cs_complete_rk45_simulation", "cs_robot_model_rk45", "cs_copy_worlds_masked_status",
    "cs_collision_reward_gym", "cs_step_observe", "cs_copy_worlds_masked_observe", "cs_refill_staged_worlds", "cs_consume_staged_worlds", "cs_gym_step",
    "cs_orca_default_math", "cs_gym_step_is_one_launch", "cs_gym_step_staged",
]


class CrowdstepError(RuntimeError):
    pass


class cs_gym_book(C.Structure):   # include/crowdstep.h cs_gym_book: the bookkeeping buffers of cs_collision_reward_gym
    _fields_ = [
        ("d_counter", C.c_void_p), ("d_seeds", C.c_void_p), ("d_mask", C.c_void_p), ("d_prev_mask", C.c_void_p), ("d_clock", C.c_void_p),
        ("clock_len", C.c_int32), ("auto_reset", C.c_int32),
        ("d_reward", C.c_void_p), ("d_terminated", C.c_void_p), ("d_truncated", C.c_void_p), ("d_info", C.c_void_p),
        ("seed_stride", C.c_uint32),
    ]


class cs_stage_book(C.Structure):   # include/crowdstep.h cs_stage_book: the tags of the pre-staged episodes
    _fields_ = [
        ("d_seeds", C.c_void_p), ("d_base_seed", C.c_void_p), ("d_epoch", C.c_void_p), ("d_staged_seed", C.c_void_p),
        ("d_staged_status", C.c_void_p), ("d_failed", C.c_void_p), ("seed_stride", C.c_uint32), ("depth", C.c_int32), ("d_pending", C.c_void_p),
    ]


class cs_worlds(C.Structure):
    _fields_ = [
        ("W", C.c_int32), ("n", C.c_int32), ("G", C.c_int32), ("O", C.c_int32), ("Smax", C.c_int32),
        ("type", C.c_int32), ("flags", C.c_int32), ("layout", C.c_int32),
        ("d_state", C.c_void_p), ("d_goals", C.c_void_p), ("d_params", C.c_void_p), ("d_safety", C.c_void_p),
        ("d_obstacles", C.c_void_p), ("d_robot", C.c_void_p), ("d_world_flags", C.c_void_p),
        ("respawn_bound_x", C.c_float), ("respawn_bound_y", C.c_float),
        ("orca_neighbor_dist", C.c_float), ("orca_time_horizon", C.c_float), ("orca_time_horizon_obst", C.c_float),
        ("orca_max_neighbors", C.c_int32), ("sm_n_actions", C.c_int32),
        ("d_orca_vertices", C.c_void_p), ("orca_n_vertices", C.c_int32),
        ("d_orca_agent_params", C.c_void_p),
        ("orca_math", C.c_int32),
    ]


_lib = None
hip_runtime_path = None  # the libamdhip64 this process uses (None: whatever the dynamic loader resolves, i.e. /opt/rocm)


def _elf_dynamic(path: str):
    """(SONAME, [NEEDED ...]) of a 64-bit little-endian ELF shared object, read from its dynamic section (no external tool)."""
    import struct

    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2 or data[5] != 1:
        return None, []
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
    sections = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    soname, needed = None, []
    for (_, typ, _, _, off, size, link, _, _, entsize) in sections:
        if typ != 6:        # SHT_DYNAMIC
            continue
        stroff = sections[link][4]
        cstr = lambda o: data[stroff + o:data.index(b"\0", stroff + o)].decode()
        for k in range(size // (entsize or 16)):
            tag, val = struct.unpack_from("<qQ", data, off + k * 16)
            if tag == 1:
                needed.append(cstr(val))
            elif tag == 14:
                soname = cstr(val)
    return soname, needed


def _preload_hip_runtime() -> None:
    """ONE HIP runtime per process, whatever the import order.  PyTorch-ROCm bundles its own libamdhip64.so (+ HSA
    runtime) under torch/lib, with the same SONAME (libamdhip64.so.7) as the system one libcrowdstep.so is linked
    against.  If libcrowdstep.so came first and pulled /opt/rocm's copy, a later `import torch` would load the bundled
    one as well: two runtimes, two device contexts, pointers of one invalid in the other.  So when torch is installed
    its bundled runtime is loaded first (RTLD_GLOBAL): the loader then binds libcrowdstep.so's NEEDED libamdhip64.so.7
    to it by SONAME, and torch, whenever it is imported, finds the very same file already mapped.
    CROWDSTEP_HIP_RUNTIME = "system" (never preload) | "torch" (default when torch is installed) | a path."""
    global hip_runtime_path
    choice = os.environ.get("CROWDSTEP_HIP_RUNTIME", "torch")
    if choice == "system":
        return
    path = choice if os.path.sep in choice else None
    if path is None:
        import importlib.util
        import sys

        if "torch" in sys.modules:   # already imported: its runtime is mapped, the SONAME match does the rest
            return
        try:
            spec = importlib.util.find_spec("torch")   # locates the package without importing it
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if not os.path.exists(cand):
            return
        # preload only when the loader would really bind libcrowdstep.so to it: torch's copy must carry the SONAME libcrowdstep.so
        # NEEDs (e.g. libamdhip64.so.7).  A ROCm-6 wheel beside a ROCm-7 system runtime has another SONAME: preloading it would map
        # two runtimes where there was one -- fall back to the system runtime and say so.
        try:
            soname, _ = _elf_dynamic(cand)
            _, needed = _elf_dynamic(LIB_PATH)
            want = [x for x in needed if x.startswith("libamdhip64")]
        except Exception:
            soname, want = None, []
        if want and soname not in want:
            import warnings

            warnings.warn(f"torch bundles {soname or 'an unreadable libamdhip64'}, libcrowdstep.so needs {want[0]}: not preloading torch's HIP runtime "
                          "(CROWDSTEP_HIP_RUNTIME=system); tensors of the two runtimes cannot be shared")
            return
        path = cand
    C.CDLL(path, mode=C.RTLD_GLOBAL)
    hip_runtime_path = path


def mapped_hip_runtimes() -> list:
    """The libamdhip64 files mapped into this process (/proc/self/maps): exactly one when the preload did its job."""
    seen = []
    try:
        for line in open("/proc/self/maps"):
            f = line.split()[-1]
            if "libamdhip64" in f and f not in seen:
                seen.append(f)
    except OSError:
        pass
    return seen


def load():
    """Load the shared library (no GPU needed for this); raises if it was never built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrowdstepError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the crowd stepper.")
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    fn = getattr(lib, "cs_abi_version", None)     # (a stale or foreign CROWDSTEP_LIB may lack the symbol: report it as ABI None, not as a ctypes AttributeError)
    got = None
    if fn is not None:
        fn.restype = C.c_int
        got = fn()
    if got != ABI_VERSION:   # a stale or foreign build (CROWDSTEP_LIB): its structs and argument lists are not the ones bound here
        raise CrowdstepError(f"{LIB_PATH} speaks ABI {got}, this binding ABI {ABI_VERSION} (include/crowdstep.h CS_ABI_VERSION): rebuild the library")
    lib.cs_last_error.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        if name != "cs_last_error" and hasattr(lib, name):
            getattr(lib, name).restype = C.c_int
    if hasattr(lib, "cs_generate_scratch_bytes"):
        lib.cs_generate_scratch_bytes.restype = C.c_size_t
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc == CS_OK:
        return
    msg = load().cs_last_error().decode(errors="replace")
    if rc == CS_ERR_TYPE:
        raise ValueError(msg)  # the reference raises ValueError for a bad type (forces_parallel.py:211)
    if rc == CS_ERR_ARG:
        raise ValueError(f"crowdstep: {msg}")
    raise CrowdstepError(f"crowdstep rc={rc}: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    rc = load().cs_device_count(C.byref(n))
    return n.value if rc == CS_OK else 0


def require_gpu() -> None:
    if device_count() < 1:
        raise CrowdstepError("no MI355X / HIP device visible: the crowd stepper has no CPU path")


def set_device(device: int) -> None:
    check(load().cs_set_device(C.c_int(device)))


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(load().cs_device_name(C.c_int(device), buf, C.c_size_t(256)))
    return buf.value.decode()


def device_pci_bus_id(device: int = 0) -> str:
    buf = C.create_string_buffer(64)
    check(load().cs_device_pci_bus_id(C.c_int(device), buf, C.c_size_t(64)))
    return buf.value.decode()


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through the C ABI (numpy in / numpy out, no torch needed)."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(load().cs_malloc(C.byref(p), C.c_size_t(max(self.nbytes, 4))))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr, dtype=np.float32, stream=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        buf = cls(arr.shape, dtype)
        buf.upload(arr, stream)
        return buf

    def upload(self, arr, stream=None):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.nbytes != self.nbytes:
            raise ValueError(f"upload size mismatch {arr.shape} vs {self.shape}")
        check(load().cs_memcpy_h2d(C.c_void_p(self.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(self.nbytes),
                                   C.c_void_p(stream)))
        if stream:
            check(load().cs_stream_sync(C.c_void_p(stream)))

    def download(self, stream=None) -> np.ndarray:
        out = np.empty(self.shape, dtype=self.dtype)
        check(load().cs_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(self.nbytes),
                                   C.c_void_p(stream)))
        return out

    @property
    def __cuda_array_interface__(self):
        """Zero-copy hand-over to torch / cupy (``torch.as_tensor(buf, device="cuda")``): the learner reads observations
        and writes actions in HBM, nothing crosses PCIe."""
        return {"shape": tuple(self.shape), "typestr": self.dtype.str, "data": (int(self.ptr), False), "version": 2,
                "strides": None}

    def torch(self):
        """A torch tensor aliasing this buffer (same HBM; the buffer must outlive the tensor)."""
        import torch

        t = torch.as_tensor(self, device="cuda")
        t._crowdstep_owner = self  # keep the allocation alive as long as the view
        return t

    def free(self):
        if getattr(self, "ptr", None):
            try:
                load().cs_free(C.c_void_p(self.ptr))
            finally:
                self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        check(load().cs_event_create(C.byref(p)))
        self.ptr = p.value

    def record(self, stream=None):
        check(load().cs_event_record(C.c_void_p(self.ptr), C.c_void_p(stream)))

    def wait(self, stream=None):
        """Make later work of `stream` wait for this event (device-side, the host does not block)."""
        check(load().cs_stream_wait_event(C.c_void_p(stream), C.c_void_p(self.ptr)))

    def done(self) -> bool:
        """True when the work recorded before this event has finished (never blocks)."""
        d = C.c_int(0)
        check(load().cs_event_query(C.c_void_p(self.ptr), C.byref(d)))
        return bool(d.value)

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float(0)
        check(load().cs_event_elapsed_ms(C.c_void_p(self.ptr), C.c_void_p(stop.ptr), C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                load().cs_event_destroy(C.c_void_p(self.ptr))
        except Exception:
            pass


class Graph:
    """with Graph.capture(stream) as g: ...launches...   then g.launch()"""

    def __init__(self, stream):
        self.stream, self.exec = stream, None

    @classmethod
    def capture(cls, stream):
        return cls(stream)

    def __enter__(self):
        check(load().cs_graph_begin_capture(C.c_void_p(self.stream)))
        return self

    def __exit__(self, exc_type, exc, tb):
        p = C.c_void_p()
        rc = load().cs_graph_end_capture(C.c_void_p(self.stream), C.byref(p))
        if exc_type is None:
            check(rc)
            self.exec = p.value
        return False

    def launch(self):
        check(load().cs_graph_launch(C.c_void_p(self.exec), C.c_void_p(self.stream)))

    def __del__(self):
        try:
            if self.exec:
                load().cs_graph_destroy(C.c_void_p(self.exec))
        except Exception:
            pass


def stream_create(priority: int = 0) -> int:
    """A non-blocking HIP stream; priority < 0: the device's highest, > 0: its lowest (background work)."""
    p = C.c_void_p()
    if priority:
        check(load().cs_stream_create_with_priority(C.byref(p), C.c_int(priority)))
    else:
        check(load().cs_stream_create(C.byref(p)))
    return p.value


def stream_sync(stream=None) -> None:
    check(load().cs_stream_sync(C.c_void_p(stream)))


def stream_destroy(stream) -> None:
    check(load().cs_stream_destroy(C.c_void_p(stream)))
