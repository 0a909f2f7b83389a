"""ctypes binding of the C ABI declared in include/mppi_gpu_amd.h.

Loads mppi_gpu_amd/lib/libmppi_gpu_amd.so (built in-tree by `__graft_entry__.build()` or
`make -C mppi_gpu_amd/csrc`).  There is no fallback of any kind: if the library is missing
this module raises at import of the symbol table, and if no HIP device is usable
`mppi_create` returns MPPI_ENODEV, which the wrappers turn into an exception.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPPI_GPU_AMD_LIB: analysis builds of the same library (tools/trace_regions.py), never a fallback
LIB_PATH = os.environ.get("MPPI_GPU_AMD_LIB") or os.path.join(_HERE, "lib", "libmppi_gpu_amd.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
engine_p = C.c_void_p

# name -> (restype, argtypes); one row per declaration in include/mppi_gpu_amd.h
SIGNATURES = {
    "mppi_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int,
                              C.POINTER(engine_p)]),
    "mppi_create_shard": (C.c_int, [C.c_int, C.c_longlong, C.c_int, C.c_float, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(engine_p)]),
    "mppi_destroy": (None, [engine_p]),
    "mppi_set_data": (C.c_int, [engine_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "mppi_set_x": (C.c_int, [engine_p, c_float_p]),
    "mppi_get_x": (C.c_int, [engine_p, c_float_p]),
    "mppi_get_act": (C.c_int, [engine_p, c_float_p]),
    "mppi_get_u": (C.c_int, [engine_p, c_float_p]),
    "mppi_get_data": (C.c_int, [engine_p, c_float_p, c_float_p]),
    "mppi_get_inf": (C.c_int, [engine_p] + [c_float_p] * 7),
    "mppi_set_params": (C.c_int, [engine_p, C.c_float, c_float_p, c_float_p]),
    "mppi_set_seed": (C.c_int, [engine_p, C.c_ulonglong]),
    "mppi_set_noise": (C.c_int, [engine_p, c_float_p]),
    "mppi_set_noise_prefetch": (C.c_int, [engine_p, C.c_int]),
    "mppi_get_prefetch_counts": (C.c_int, [engine_p, C.POINTER(C.c_longlong)]),
    "mppi_set_ref_compat": (C.c_int, [engine_p, C.c_int]),
    "mppi_set_noise_store": (C.c_int, [engine_p, C.c_int]),
    "mppi_set_action_limit": (C.c_int, [engine_p, c_float_p]),
    "mppi_set_tuning": (C.c_int, [engine_p, C.c_int, C.c_int, C.c_int]),
    "mppi_set_pipeline": (C.c_int, [engine_p, C.c_int]),
    "mppi_get_pipeline": (C.c_int, [engine_p, c_int_p, c_int_p]),
    "mppi_set_packing": (C.c_int, [engine_p, C.c_int]),
    "mppi_get_layout": (C.c_int, [engine_p, c_int_p]),
    "mppi_solve_async": (C.c_int, [engine_p, C.c_void_p]),
    "mppi_flush_async": (C.c_int, [engine_p]),
    "mppi_sync_act": (C.c_int, [engine_p, c_float_p]),
    "mppi_wait_act": (C.c_int, [engine_p, c_float_p]),
    "mppi_partial_len": (C.c_int, [engine_p]),
    "mppi_solve_local_async": (C.c_int, [engine_p, C.c_void_p, C.c_void_p]),
    "mppi_solve_finish_async": (C.c_int, [engine_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mppi_xchg_handle_bytes": (C.c_int, []),
    "mppi_xchg_open": (C.c_int, [engine_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_xchg_connect": (C.c_int, [engine_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_xchg_set_timeout": (C.c_int, [engine_p, C.c_double]),
    "mppi_solve_exchange_async": (C.c_int, [engine_p, C.c_void_p]),
    "mppi_xchg_close": (C.c_int, [engine_p]),
    "mppi_set_profiling": (C.c_int, [engine_p, C.c_int]),
    "mppi_kernel_ms": (C.c_int, [engine_p, C.c_int, c_double_p, c_int_p]),
    "mppi_get_geometry": (C.c_int, [engine_p, c_int_p]),
    "mppi_get_launch_counts": (C.c_int, [engine_p, C.POINTER(C.c_longlong)]),
    "mppi_cpu_create": (C.c_void_p, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int]),
    "mppi_cpu_destroy": (None, [C.c_void_p]),
    "mppi_cpu_set_data": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p]),
    "mppi_cpu_set_params": (C.c_int, [C.c_void_p, C.c_float, c_float_p, c_float_p]),
    "mppi_cpu_set_seed": (C.c_int, [C.c_void_p, C.c_ulonglong]),
    "mppi_cpu_set_noise": (C.c_int, [C.c_void_p, c_float_p]),
    "mppi_cpu_set_threads": (C.c_int, [C.c_void_p, C.c_int]),
    "mppi_cpu_next": (C.c_int, [C.c_void_p, c_float_p, c_float_p]),
    "mppi_cpu_get": (C.c_int, [C.c_void_p] + [c_float_p] * 6),
    "mppi_device_count": (C.c_int, []),
    "mppi_last_error": (C.c_char_p, []),
    "mppi_version": (C.c_char_p, []),
}

# include/mppi_gpu_amd_sharded.h (libmppi_gpu_amd_sharded.so: the single-process multi-GPU host)
sharded_p = C.c_void_p
SHARDED_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libmppi_gpu_amd_sharded.so")
SHARDED_SIGNATURES = {
    "mppi_sharded_create": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                      c_int_p, C.c_int, C.POINTER(sharded_p)]),
    "mppi_sharded_destroy": (None, [sharded_p]),
    "mppi_sharded_set_data": (C.c_int, [sharded_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "mppi_sharded_set_x": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_get_x": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_get_u": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_get_act": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_solve_async": (C.c_int, [sharded_p]),
    "mppi_sharded_sync_act": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_get_inf": (C.c_int, [sharded_p] + [c_float_p] * 7),
    "mppi_sharded_get_data": (C.c_int, [sharded_p, c_float_p, c_float_p]),
    "mppi_sharded_set_params": (C.c_int, [sharded_p, C.c_float, c_float_p, c_float_p]),
    "mppi_sharded_set_seed": (C.c_int, [sharded_p, C.c_ulonglong]),
    "mppi_sharded_set_noise": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_set_action_limit": (C.c_int, [sharded_p, c_float_p]),
    "mppi_sharded_set_timeout": (C.c_int, [sharded_p, C.c_double]),
    "mppi_sharded_n_shards": (C.c_int, [sharded_p]),
    "mppi_sharded_transport": (C.c_int, [sharded_p]),
    "mppi_sharded_shard_info": (C.c_int, [sharded_p, C.c_int, C.POINTER(C.c_longlong)]),
    "mppi_sharded_engine": (engine_p, [sharded_p, C.c_int]),
    "mppi_sharded_last_error": (C.c_char_p, []),
}

_lib = None
_slib = None


def load():
    """Return the loaded library with prototypes set; raise if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own libamdhip64.so with the SAME soname as /opt/rocm's.  Whichever is
    # loaded first serves every later DT_NEEDED of that soname, so torch must come first: then
    # this library, torch's allocator/streams and RCCL all share ONE HIP runtime (stream and
    # device-pointer handles are interchangeable).  The other order leaves two runtimes in the
    # process and torch cannot see the GPU.  Without torch the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C mppi_gpu_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_sharded():
    """The single-process multi-GPU host library (links libmppi_gpu_amd.so and librccl)."""
    global _slib
    if _slib is not None:
        return _slib
    load()      # torch (its HIP runtime and RCCL) first, then the engine library
    if not os.path.exists(SHARDED_LIB_PATH):
        raise ImportError(f"{SHARDED_LIB_PATH} is missing: build it with `make -C mppi_gpu_amd/csrc`")
    lib = C.CDLL(SHARDED_LIB_PATH)
    for name, (res, args) in SHARDED_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _slib = lib
    return lib


class MppiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mppi_gpu_amd error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != 0:
        raise MppiError(rc, load().mppi_last_error().decode(errors="replace"))
