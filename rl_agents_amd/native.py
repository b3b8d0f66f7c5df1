"""ctypes binding of libmi355plan.so (include/mi355plan.h) -- the only compute path of this package.

There is no CPU fallback: importing works anywhere (so that host logic can be unit-tested), but
creating a :class:`Context` raises if the library is missing or no MI355X is visible.
north_star asks for a cffi ABI-mode layer; cffi is not installed in this image, ctypes binds the
same C ABI (INTEGRATION.md shows both).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

MP_MEM_HOST, MP_MEM_DEVICE, MP_MEM_RNG_DEVICE = 0, 1, 2
MP_OK, MP_ERR_HIP, MP_ERR_REWARD_RANGE, MP_ERR_ALLOC, MP_ERR_ARG, MP_ERR_MODE = 0, -1, -2, -3, -4, -5
MODE_DETERMINISTIC, MODE_STOCHASTIC, MODE_SPARSE, MODE_CARTPOLE = 0, 1, 2, 3
ERR_REWARD_RANGE, ERR_ARG, ERR_MODE = -2, -4, -5

_LIB = None

c_i32, c_i64, c_f64, c_u8, c_u64 = C.c_int32, C.c_int64, C.c_double, C.c_uint8, C.c_uint64
P = C.POINTER
_vp = C.c_void_p

# every symbol include/mi355plan.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mp_last_error": (C.c_char_p, []),
    "mp_abi_version": (C.c_int, []),
    "mp_ctx_create": (C.c_int, [C.c_int, _vp, P(_vp)]),
    "mp_ctx_destroy": (C.c_int, [_vp]),
    "mp_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "mp_ctx_synchronize": (C.c_int, [_vp]),
    "mp_ctx_device_faults": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "mp_ctx_get_stream": (C.c_int, [_vp, P(_vp)]),
    "mp_ctx_device_info": (C.c_int, [_vp, P(c_i32), P(c_i32), P(c_i64), P(c_i64), C.c_char_p, c_i32]),
    "mp_model_load_table": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, c_i32, P(_vp)]),
    "mp_model_load_table_batch": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, c_i32, P(_vp)]),
    "mp_model_batch_info": (C.c_int, [_vp, P(c_i32), P(c_i32)]),
    "mp_model_update_tables": (C.c_int, [_vp, c_i32, c_i32, _vp, _vp, _vp]),
    "mp_model_update_rows": (C.c_int, [_vp, c_i32, _vp, _vp, _vp, _vp]),
    "mp_vi_solve_batch": (C.c_int, [_vp, _vp, c_f64, c_i32, c_f64, c_f64, _vp, _vp, c_i32]),
    "mp_uct_plan_models": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, _vp, c_i32, c_i32, c_f64, c_f64, _vp, _vp, _vp, c_i32, _vp, _vp,
                                     _vp, _vp, _vp, _vp, c_i32]),
    "mp_opd_plan_models": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, c_i32, c_f64, c_f64, _vp, c_i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                     c_i32]),
    "mp_comm_unique_id": (C.c_int, [_vp]),
    "mp_comm_init": (C.c_int, [_vp, c_i32, c_i32, _vp]),
    "mp_gather_results": (C.c_int, [_vp, _vp, c_i32, c_i32, _vp, _vp]),
    "mp_comm_destroy": (C.c_int, [_vp]),
    "mp_libm_sincos_variant": (C.c_int, []),
    "mp_libm_sincos": (C.c_int, [c_i32, _vp, c_i32, _vp, _vp]),
    "mp_selftest_sincos": (C.c_int, [_vp, c_i32, _vp, c_i32, _vp, _vp]),
    "mp_model_load_dense": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, P(_vp)]),
    "mp_model_load_dense_rows": (C.c_int, [_vp, c_i32, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, P(_vp)]),
    "mp_vi_backup": (C.c_int, [_vp, _vp, c_f64, c_i32, _vp, _vp, c_i32]),
    "mp_model_load_sparse": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, _vp, P(_vp)]),
    "mp_model_load_cartpole": (C.c_int, [_vp, _vp, P(_vp)]),
    "mp_model_free": (C.c_int, [_vp]),
    "mp_model_info": (C.c_int, [_vp, P(c_i32), P(c_i32), P(c_i32), P(c_i32), P(c_i32)]),
    "mp_vi_solve": (C.c_int, [_vp, _vp, c_f64, c_i32, c_f64, c_f64, c_i32, _vp, _vp, c_i32]),
    "mp_vi_solve_v": (C.c_int, [_vp, _vp, c_f64, c_i32, c_f64, c_f64, _vp, c_i32]),
    "mp_vi_solve_v_robust": (C.c_int, [_vp, _vp, c_f64, c_i32, c_f64, c_f64, _vp, c_i32]),
    "mp_vi_sweeps": (C.c_int, [_vp, _vp, c_f64, c_i32, c_i32]),
    "mp_vi_dense_mode": (C.c_int, [_vp, c_i32]),
    "mp_uct_record_visits": (C.c_int, [_vp, _vp]),
    "mp_vi_exact_plan": (C.c_int, [c_i32, c_i32, _vp, c_i32, _vp, c_i32, _vp, _vp]),
    "mp_uct_plan": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, c_i32, c_i32, c_f64, c_f64, _vp, _vp, _vp, c_i32, _vp, _vp,
                              _vp, _vp, _vp, _vp, c_i32]),
    "mp_policy_load": (C.c_int, [_vp, _vp, _vp, _vp, P(_vp)]),
    "mp_policy_free": (C.c_int, [_vp]),
    "mp_uct_plan_policy": (C.c_int, [_vp, _vp, _vp, c_i32, _vp, _vp, c_i32, c_i32, c_f64, c_f64, _vp, c_i32, _vp, _vp,
                                     _vp, _vp, _vp, _vp, c_i32]),
    "mp_uct_step_tree": (C.c_int, [_vp, c_i32, _vp, c_i32]),
    "mp_uct_reset_tree": (C.c_int, [_vp]),
    "mp_uct_tree_capacity": (C.c_int, [_vp, P(c_i32)]),
    "mp_uct_tree_export": (C.c_int, [_vp, c_i32, c_i32, P(c_i32), _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_uct_path_count": (C.c_int, [_vp, c_i32, _vp, c_i32, P(C.c_int64)]),
    "mp_model_set_available": (C.c_int, [_vp, _vp]),
    "mp_model_set_episode_rules": (C.c_int, [_vp, c_i32, c_i32]),
    "mp_uct_plan_stochastic": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, c_i32, c_i32, c_f64, c_f64, _vp, _vp, c_i32, _vp, _vp,
                                         c_i32, _vp, _vp, _vp, _vp, _vp, _vp, c_i32]),
    "mp_uct_plan_stochastic_policy": (C.c_int, [_vp, _vp, _vp, c_i32, _vp, _vp, c_i32, c_i32, c_f64, c_f64, c_i32, _vp, _vp,
                                                c_i32, _vp, _vp, _vp, _vp, _vp, _vp, c_i32]),
    "mp_uct_stoch_tree_priors": (C.c_int, [_vp, c_i32, _vp]),
    "mp_uct_stoch_tree_capacity": (C.c_int, [_vp, P(c_i32)]),
    "mp_uct_stoch_tree_export": (C.c_int, [_vp, c_i32, c_i32, P(c_i32), _vp, _vp, _vp, _vp, _vp]),
    "mp_policy_load_listed": (C.c_int, [_vp, _vp, _vp, _vp, _vp, P(_vp)]),
    "mp_policy_load_ordered": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, P(_vp)]),
    "mp_policy_load_rows": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, c_i32, P(_vp)]),
    "mp_opd_plan": (C.c_int, [_vp, _vp, c_i32, _vp, c_i32, c_f64, c_f64, _vp, c_i32, _vp, _vp, _vp, _vp, _vp, _vp,
                              c_i32]),
    "mp_opd_tree_export": (C.c_int, [_vp, c_i32, c_i32, P(c_i32), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_model_load_joint": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, P(_vp)]),
    "mp_ropd_plan": (C.c_int, [_vp, _vp, c_i32, _vp, c_i32, c_f64, c_f64, _vp, c_i32, _vp, _vp, _vp, _vp, _vp, _vp, c_i32]),
    "mp_ropd_tree_export": (C.c_int, [_vp, c_i32, c_i32, P(c_i32), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_model_set_available_joint": (C.c_int, [_vp, _vp]),
    "mp_saopd_create": (C.c_int, [_vp, _vp, c_i32, P(_vp)]),
    "mp_saopd_free": (C.c_int, [_vp]),
    "mp_saopd_plan": (C.c_int, [_vp, _vp, _vp, c_i32, c_f64, c_f64, c_f64, c_i32, c_i32, _vp, c_i32, _vp, _vp, _vp, _vp,
                                _vp, c_i32]),
    "mp_saopd_info": (C.c_int, [_vp, P(c_i32), P(c_i32), P(c_i32), P(c_i32)]),
    "mp_saopd_export": (C.c_int, [_vp, c_i32, c_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mp_olop_allocation": (C.c_int, [c_i32, c_f64, P(c_i32), P(c_i32)]),
    "mp_last_kernel_ms": (C.c_int, [_vp, P(c_f64), P(c_i32)]),
    "mp_last_kernel_variant": (C.c_char_p, [_vp]),
    "mp_selftest_lds_atomic_order": (C.c_int, [_vp, C.c_int32, C.POINTER(C.c_int64)]),
    "mp_env_step": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, _vp, _vp, c_i32, c_i32, _vp, _vp, _vp, _vp, c_i32, _vp, c_i32]),
    "mp_greedy_actions": (C.c_int, [_vp, c_i32, c_i32, c_i32, _vp, _vp, _vp, c_i32, c_i32]),
    "mp_env_step_stochastic": (C.c_int, [_vp, _vp, c_i32, _vp, _vp, _vp, _vp, c_i32, c_i32, _vp, _vp, _vp, _vp, c_i32, _vp, _vp,
                                         c_i32]),
    "mp_host_alloc": (C.c_int, [_vp, c_i64, P(_vp)]),
    "mp_host_free": (C.c_int, [_vp, _vp]),
    "mp_rng_create": (C.c_int, [_vp, c_i32, P(_vp)]),
    "mp_rng_free": (C.c_int, [_vp]),
    "mp_rng_set": (C.c_int, [_vp, c_i32, c_i32, _vp]),
    "mp_rng_get": (C.c_int, [_vp, c_i32, c_i32, _vp]),
    "mp_rng_seed_sequence": (C.c_int, [_vp, c_i32, c_i32, _vp, c_i32, c_i64]),
    "mp_rng_device_ptr": (_vp, [_vp, c_i32]),
    "mp_seed_sequence_states": (C.c_int, [_vp, c_i32, c_i64, c_i32, _vp]),
    "mp_pack_rows": (C.c_int, [_vp, _vp, c_i32, c_i32, c_i32, _vp, _vp, _vp]),
    "mp_unpack_rows": (C.c_int, [_vp, _vp, c_i32, c_i32, c_i32, c_i32, _vp, _vp, _vp]),
}


class CartPoleParams(C.Structure):
    _fields_ = [("gravity", c_f64), ("masscart", c_f64), ("masspole", c_f64), ("length", c_f64),
                ("force_mag", c_f64), ("tau", c_f64), ("theta_threshold", c_f64), ("x_threshold", c_f64),
                ("max_steps", c_i32), ("euler", c_i32)]


class NativeError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libmi355plan error {}: {}".format(code, message))
        self.code = code


def lib_path():
    """The in-tree library; MI355PLAN_LIB points at another build of the same sources (kernel A/B experiments)."""
    return os.environ.get("MI355PLAN_LIB") or _build.LIB_PATH


def _bind_torch_hip_runtime():
    """One HIP runtime per process: when PyTorch is installed, libmi355plan.so must bind to the copy of libamdhip64 that torch
    ships (streams, events and device pointers then cross freely between the two).  Rounds 1-5 got that by importing torch before
    loading the library -- 0.55 s of the first act() of a process that never uses torch (profiles/r06_first_act.txt).  Now: if
    torch is already imported, nothing to do; else torch's libamdhip64 is loaded BY PATH (importlib finds the package without
    importing it), so the library binds to it and a later `import torch` finds its runtime already in the process.
    MI355PLAN_EAGER_TORCH=1 restores the import."""
    import sys
    if "torch" in sys.modules:
        return
    if os.environ.get("MI355PLAN_EAGER_TORCH"):
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the C ABI itself
            pass
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        for root in (list(spec.submodule_search_locations) if spec is not None and spec.submodule_search_locations else []):
            cand = os.path.join(root, "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
                return
    except Exception:  # pragma: no cover - no torch, or an unusual layout: the library then loads the system runtime
        pass


def load():
    """Load libmi355plan.so; raises (never falls back) if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError("{} is missing: run `python -m rl_agents_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the planning kernels.".format(path))
    if not os.environ.get("MI355PLAN_NO_TORCH"):
        _bind_torch_hip_runtime()
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.mp_abi_version() != 7:
        raise RuntimeError("libmi355plan ABI version mismatch")
    _LIB = lib
    return lib


def _check(rc):
    if rc != 0:
        raise NativeError(rc, load().mp_last_error().decode("utf-8", "replace"))


def _ptr(a):
    """numpy array / torch tensor / None -> void* address."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    raise TypeError("expected numpy array, torch tensor or None, got {}".format(type(a)))


def entropy_words(entropy):
    """numpy's split of entropy integers into little-endian uint32 words (bit_generator.pyx _coerce_to_uint32_array):
    an int or a sequence of ints, each non-negative; 0 -> [0]."""
    ints = [entropy] if isinstance(entropy, (int, np.integer)) else list(entropy)
    words = []
    for v in ints:
        v = int(v)
        if v < 0:
            raise ValueError("expected non-negative integer")
        if v == 0:
            words.append(0)
        while v > 0:
            words.append(v & 0xFFFFFFFF)
            v >>= 32
    return np.asarray(words, dtype=np.uint32)


def seed_sequence_states(entropy, first_key, count):
    """PCG64 records uint64 [count, 6] of Generator(PCG64(SeedSequence(list(entropy) + [first_key + i]))), i < count --
    numpy's SeedSequence hashing and PCG64 seeding restated in C (mp_seed_sequence_states: host arithmetic, no GPU),
    ~1000x faster than constructing the generators in Python (262 144 roots: seconds -> milliseconds).  ``entropy``:
    int, sequence of ints, or () for SeedSequence(first_key + i)."""
    w = entropy_words(entropy) if not (hasattr(entropy, "__len__") and len(entropy) == 0) else np.zeros(0, np.uint32)
    out = np.zeros((int(count), 6), dtype=np.uint64)
    _check(load().mp_seed_sequence_states(_ptr(w) if len(w) else None, len(w), int(first_key), int(count), _ptr(out)))
    return out


def libm_sincos_variant():
    """Which restated form of glibc's small-argument sin / cos reproduces this host's libm (mp_libm_sincos_variant):
    1 = FMA-contracted, 2 = every operation rounded, 0 = neither (CartPole then keeps the device's own sincos)."""
    return int(load().mp_libm_sincos_variant())


def libm_sincos(x, variant):
    """The restated functions on the host: (sin, cos) arrays of ``x`` in form ``variant`` (0 = libm itself)."""
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    s, c = np.zeros_like(x), np.zeros_like(x)
    _check(load().mp_libm_sincos(int(x.size), _ptr(x), int(variant), _ptr(s), _ptr(c)))
    return s, c


def olop_allocation(budget, gamma):
    """OLOP.allocation (tree_search/olop.py:50-62): budget -> (episodes, horizon)."""
    e, h = c_i32(), c_i32()
    rc = load().mp_olop_allocation(int(budget), float(gamma), C.byref(e), C.byref(h))
    if rc != 0:
        raise ValueError("Could not split budget {} with gamma {}".format(budget, gamma))
    return e.value, h.value


def vi_exact_plan(n):
    """The tables the bit-exact dense backup sums a row of ``n`` elements by (numpy's pairwise recursion; host only):
    -> (leaves int32 [L,2] {offset, length}, nodes int32 [K,2] {left slot, right slot} by height, hoff int32 [H+1],
    most 8-element steps in a leaf).  Leaf l is slot l, addition k slot L + k; the last slot holds the sum."""
    lib = load()
    counts = np.zeros(4, dtype=np.int32)
    _check(lib.mp_vi_exact_plan(int(n), 0, None, 0, None, 0, None, _ptr(counts)))
    leaves = np.zeros((int(counts[0]), 2), dtype=np.int32)
    nodes = np.zeros((int(counts[1]), 2), dtype=np.int32)
    hoff = np.zeros(int(counts[2]) + 1, dtype=np.int32)
    _check(lib.mp_vi_exact_plan(int(n), len(leaves), _ptr(leaves), len(nodes), _ptr(nodes) if len(nodes) else None,
                                len(hoff) - 1, _ptr(hoff), _ptr(counts)))
    return leaves, nodes, hoff, int(counts[3])


class Context(object):
    """One GPU + stream + workspaces (mp_ctx).  `stream` = raw hipStream_t (int) or None."""

    def __init__(self, device=0, stream=None):
        self._lib = load()
        h = _vp()
        _check(self._lib.mp_ctx_create(int(device), _vp(stream) if stream else None, C.byref(h)))
        self._h = h
        self.device = int(device)

    @classmethod
    def on_torch_stream(cls, device=0):
        """Context enqueuing on a torch side stream of `device` (kept as ``ctx.torch_stream``): run torch work
        under ``with torch.cuda.stream(ctx.torch_stream)`` and torch events / allocations interoperate."""
        import torch
        with torch.cuda.device(device):
            stream = torch.cuda.Stream(device=device)
            ctx = cls(device, stream.cuda_stream)
            ctx.torch_stream = stream
            return ctx

    def close(self):
        if getattr(self, "_h", None):
            for buf in self.__dict__.pop("_pin_cache", {}).values():
                buf.close()
            self._lib.mp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        """Wait for the context's stream; raises NativeError(MP_ERR_ARG) if a device-array plan since the last synchronisation
        had to clamp out-of-range roots (mp_ctx_synchronize)."""
        _check(self._lib.mp_ctx_synchronize(self._h))

    def device_faults(self):
        """Roots clamped by device-array plans on batch models since the last synchronize() (exact once the work was waited for)."""
        n = c_i32()
        _check(self._lib.mp_ctx_device_faults(self._h, C.byref(n)))
        return int(n.value)

    def set_stream(self, stream):
        """Enqueue on another hipStream_t from now on (raw pointer, e.g. ``torch.cuda.current_stream().cuda_stream``)."""
        _check(self._lib.mp_ctx_set_stream(self._h, _vp(stream) if stream else None))

    def stream_ptr(self):
        """The raw hipStream_t this context enqueues on (``torch.cuda.ExternalStream(ptr)`` wraps it)."""
        st = _vp()
        _check(self._lib.mp_ctx_get_stream(self._h, C.byref(st)))
        return st.value or 0

    def device_info(self):
        cu, wave, lds, hbm = c_i32(), c_i32(), c_i64(), c_i64()
        name = C.create_string_buffer(256)
        _check(self._lib.mp_ctx_device_info(self._h, C.byref(cu), C.byref(wave), C.byref(lds), C.byref(hbm), name, 256))
        return dict(n_cu=cu.value, wave_size=wave.value, lds_bytes=lds.value, hbm_bytes=hbm.value,
                    name=name.value.decode())

    def last_kernel_ms(self):
        ms, n = c_f64(), c_i32()
        _check(self._lib.mp_last_kernel_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def last_kernel_variant(self):
        """Which kernel variant the last UCT plan launched ("uct_global", "uct_ldsr", ...)."""
        return self._lib.mp_last_kernel_variant(self._h).decode()

    # ---- the collective of the C ABI (RCCL resolved at run time): what a consumer without torch.distributed calls ------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL unique id (rank 0 creates it, the caller distributes it)."""
        uid = np.zeros(128, dtype=np.uint8)
        _check(load().mp_comm_unique_id(_ptr(uid)))
        return uid

    def comm_init(self, rank, world, uid):
        uid = np.ascontiguousarray(uid, dtype=np.uint8).reshape(128)
        _check(self._lib.mp_comm_init(self._h, int(rank), int(world), _ptr(uid)))

    def gather_results(self, packed, gathered, stream=None):
        """ncclAllGather of this rank's packed rows (uint8 device tensor [per, row_bytes]) into ``gathered`` [world * per, row_bytes]."""
        _check(self._lib.mp_gather_results(self._h, _vp(stream) if stream else None, int(packed.shape[0]), int(packed.shape[1]),
                                           _ptr(packed), _ptr(gathered)))

    def comm_destroy(self):
        _check(self._lib.mp_comm_destroy(self._h))

    def selftest_sincos(self, x, variant):
        """(sin, cos) of ``x`` evaluated by the DEVICE's restated libm functions in form ``variant`` (mp_selftest_sincos)."""
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
        s, c = np.zeros_like(x), np.zeros_like(x)
        _check(self._lib.mp_selftest_sincos(self._h, int(x.size), _ptr(x), int(variant), _ptr(s), _ptr(c)))
        return s, c

    def selftest_lds_atomic_order(self, waves=65536):
        """Violations of "same-address LDS atomics of one wave instruction apply in lane order" (0 on a conforming device)."""
        v = C.c_int64(-1)
        _check(self._lib.mp_selftest_lds_atomic_order(self._h, int(waves), C.byref(v)))
        return int(v.value)

    # ---- host-inclusive fast path: pinned arrays, device-resident generator records -------------------------
    def pinned(self, spec):
        """``{name: (shape, dtype)}`` -> :class:`PinnedArrays`: numpy views of ONE pinned host block (mp_host_alloc).
        Copies from / to them are asynchronous DMA; pass them where host arrays are expected."""
        return PinnedArrays(self, spec)

    def plan_buffers(self, n_roots, max_plan_len, n_actions=None, outputs=("plans", "plan_len", "root_value", "env_steps")):
        """Pinned, reusable host arrays for batched plan() calls of ``n_roots`` roots: the ``root_state`` input and the
        requested ``outputs`` (``uct_plan(..., out=buffers)`` fills exactly those; outputs not listed are not computed
        into host memory at all).  The caller owns them: a later call with the same buffers overwrites the results."""
        n, mpl = int(n_roots), int(max_plan_len)
        shapes = dict(root_state=((n,), np.int32), plans=((n, mpl), np.int32), plan_len=((n,), np.int32),
                      root_value=((n,), np.float64), env_steps=((n,), np.int64), root_lower=((n,), np.float64),
                      root_upper=((n,), np.float64), status=((n,), np.int32))
        if n_actions is not None:
            shapes.update(root_child_count=((n, int(n_actions)), np.int64), root_child_value=((n, int(n_actions)), np.float64))
        names = ["root_state"] + [k for k in outputs]
        return PinnedArrays(self, {k: shapes[k] for k in names})

    # Host-array plans of FEW roots (a single agent's act(): one root) are made through pinned scratch arrays the context
    # keeps: the kernel reads and writes them in place (zero-copy) and the call is one launch + one synchronisation instead of
    # nine pageable copies, each a driver round trip longer than the data (MCTSAgent.act 0.40 -> ms: tools/agent_latency.py).
    SMALL_PLAN_ROOTS = 64

    def _scratch(self, kind, n, spec):
        cache = self.__dict__.setdefault("_pin_cache", {})
        key = (kind, n) + tuple((k,) + tuple(v[0]) for k, v in spec.items())
        buf = cache.get(key)
        if buf is None:
            if len(cache) >= 8:                                  # a handful of shapes per process: evict the oldest
                cache.pop(next(iter(cache))).close()
            buf = cache[key] = PinnedArrays(self, spec)
        return buf

    def device_rng(self, states_or_n):
        """Generator records resident on the device (mp_rng): an int n (uninitialised) or a uint64 [n, 6] array."""
        if isinstance(states_or_n, (int, np.integer)):
            return DeviceRng(self, int(states_or_n))
        st = np.ascontiguousarray(states_or_n, dtype=np.uint64).reshape(-1, 6)
        rng = DeviceRng(self, st.shape[0])
        rng.set(st)
        return rng

    # ---- batched evaluation on the device ------------------------------------------------------------------------
    def env_step_device(self, model, state, steps, alive, plans, max_steps, gpow, returns, discounted, actions_log, n_alive):
        """One lock-step env.step of every live episode (mp_env_step): all arguments are device tensors; only enqueues."""
        n = int(state.shape[0])
        _check(self._lib.mp_env_step(self._h, model._h, n, _ptr(state), _ptr(steps), _ptr(alive), _ptr(plans),
                                     int(plans.shape[1]), int(max_steps), _ptr(gpow), _ptr(returns), _ptr(discounted),
                                     _ptr(actions_log), 0 if actions_log is None else int(actions_log.shape[1]),
                                     _ptr(n_alive), MP_MEM_DEVICE))

    def env_step_stochastic_device(self, model, state, steps, alive, plans, max_steps, gpow, returns, discounted, actions_log,
                                   n_alive, env_rng):
        """mp_env_step_stochastic: env_step_device for stochastic / sparse models; env_rng int64 [n, 6] device records of the
        episodes' own env generators, advanced in place."""
        n = int(state.shape[0])
        _check(self._lib.mp_env_step_stochastic(self._h, model._h, n, _ptr(state), _ptr(steps), _ptr(alive), _ptr(plans),
                                                int(plans.shape[1]), int(max_steps), _ptr(gpow), _ptr(returns), _ptr(discounted),
                                                _ptr(actions_log), 0 if actions_log is None else int(actions_log.shape[1]),
                                                _ptr(n_alive), _ptr(env_rng), MP_MEM_DEVICE))

    def uct_plan_stochastic_device(self, model, n_roots, root_state, episodes, horizon, gamma, temperature, prior_p, rollout_p,
                                   rng_state, env_rng_state, max_plan_len, closed_loop=False, plans=None, plan_len=None,
                                   root_value=None, env_steps=None, root_steps=None, policy=None):
        """mp_uct_plan_stochastic / _policy on device tensors; only enqueues."""
        if policy is not None:
            _check(self._lib.mp_uct_plan_stochastic_policy(self._h, model._h, policy._h, int(n_roots), _ptr(root_state),
                                                           _ptr(root_steps), int(episodes), int(horizon), float(gamma),
                                                           float(temperature), int(bool(closed_loop)), _ptr(rng_state),
                                                           _ptr(env_rng_state), int(max_plan_len), _ptr(plans), _ptr(plan_len),
                                                           _ptr(root_value), None, None, _ptr(env_steps), MP_MEM_DEVICE))
            return
        pp = np.ascontiguousarray(prior_p, dtype=np.float64)
        rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
        _check(self._lib.mp_uct_plan_stochastic(self._h, model._h, int(n_roots), _ptr(root_state), _ptr(root_steps), int(episodes),
                                                int(horizon), float(gamma), float(temperature), _ptr(pp), _ptr(rp),
                                                int(bool(closed_loop)), _ptr(rng_state), _ptr(env_rng_state), int(max_plan_len),
                                                _ptr(plans), _ptr(plan_len), _ptr(root_value), None, None, _ptr(env_steps),
                                                MP_MEM_DEVICE))

    def greedy_actions_device(self, q, state, plans):
        """plans[:, 0] = argmax_a q[state, a] (first maximum), device tensors."""
        _check(self._lib.mp_greedy_actions(self._h, int(state.shape[0]), int(q.shape[0]), int(q.shape[1]), _ptr(q),
                                           _ptr(state), _ptr(plans), int(plans.shape[1]), MP_MEM_DEVICE))

    # ---- result exchange of the sharded path (mp_pack_rows / mp_unpack_rows): device tensors, only enqueues ------------
    @staticmethod
    def _row_args(arrays):
        ptrs = (_vp * len(arrays))(*[_ptr(a) for a in arrays])
        widths = (c_i32 * len(arrays))(*[int(a.element_size()) * int(np.prod(tuple(a.shape[1:]), dtype=np.int64))
                                         for a in arrays])
        return ptrs, widths

    def pack_rows(self, arrays, n_local, packed, stream=None):
        """Per-root device arrays [n_local, ...] -> rows of ``packed`` (uint8 [per, row_bytes]); padding rows zeroed."""
        ptrs, widths = self._row_args(arrays)
        _check(self._lib.mp_pack_rows(self._h, _vp(stream) if stream else None, int(n_local), int(packed.shape[0]),
                                      len(arrays), ptrs, widths, _ptr(packed)))

    def unpack_rows(self, gathered, n_total, world, arrays, stream=None):
        """``gathered`` (uint8 [world * per, row_bytes], rank-major) -> the full per-root arrays [n_total, ...]."""
        ptrs, widths = self._row_args(arrays)
        _check(self._lib.mp_unpack_rows(self._h, _vp(stream) if stream else None, int(n_total), int(world),
                                        int(gathered.shape[0]) // int(world), len(arrays), _ptr(gathered), widths, ptrs))

    # ---- models ------------------------------------------------------------------------------
    def load_table(self, transition, reward, terminal=None, done_rule="source", max_steps=0, available=None):
        """Deterministic tables: transition int [S,A] or [M,S,A], reward like it, terminal [S].
        available: bool [S,A], the actions state.get_available_actions() lists per state (None = all)."""
        t = np.ascontiguousarray(transition, dtype=np.int64)
        r = np.ascontiguousarray(reward, dtype=np.float64)
        if t.shape != r.shape or t.ndim not in (2, 3):
            raise ValueError("transition and reward must both be [S, A] (or [M, S, A])")
        m = 1 if t.ndim == 2 else t.shape[0]
        s, a = t.shape[-2:]
        term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(s).astype(np.uint8))
        h = _vp()
        _check(self._lib.mp_model_load_table(self._h, m, s, a, _ptr(t), _ptr(r), _ptr(term),
                                             int(done_rule == "next"), int(max_steps or 0), C.byref(h)))
        model = Model(self, h, MODE_DETERMINISTIC, m, s, a, 0)
        if available is not None:
            av = np.ascontiguousarray(np.asarray(available).reshape(s, a).astype(np.uint8))
            _check(self._lib.mp_model_set_available(h, _ptr(av)))
            model.available = av.astype(bool)
        return model

    def load_table_batch(self, transition, reward, terminal=None, done_rule="source", max_steps=0):
        """A batch of N independent deterministic MDPs of one shape (mp_model_load_table_batch): transition int [N,S,A]
        (states LOCAL to each MDP), reward [N,S,A], terminal [N,S] or None -> one :class:`Model` over the N * S GLOBAL
        states ``b * S + s`` (``model.n_models``, ``model.S_each``); every deterministic-table entry point takes it, the
        planners with ``model_index=`` (one MDP per root) or with global root states."""
        t = np.ascontiguousarray(transition, dtype=np.int64)
        r = np.ascontiguousarray(reward, dtype=np.float64)
        if t.shape != r.shape or t.ndim != 3:
            raise ValueError("transition and reward must both be [N, S, A]")
        n, s, a = t.shape
        term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(n, s).astype(np.uint8))
        h = _vp()
        _check(self._lib.mp_model_load_table_batch(self._h, n, s, a, _ptr(t), _ptr(r), _ptr(term), int(done_rule == "next"),
                                                   int(max_steps or 0), C.byref(h)))
        model = Model(self, h, MODE_DETERMINISTIC, 1, n * s, a, 0)
        model.n_models, model.S_each = n, s
        return model

    def vi_solve_batch(self, model, gamma, iterations, rtol=1e-5, atol=1e-8):
        """N value-iteration agents in one launch (mp_vi_solve_batch) -> (Q float64 [N,S,A], sweeps int32 [N]): each MDP of
        the batch model runs to its own allclose exit, bit-equal to N :meth:`vi_solve` calls on the N tables."""
        q = np.zeros((model.n_models, model.S_each, model.A), dtype=np.float64)
        sweeps = np.zeros(model.n_models, dtype=np.int32)
        _check(self._lib.mp_vi_solve_batch(self._h, model._h, float(gamma), int(iterations), float(rtol), float(atol),
                                           _ptr(q), _ptr(sweeps), MP_MEM_HOST))
        return q, sweeps

    def vi_solve_batch_device(self, model, gamma, iterations, q_out, sweeps_out, rtol=1e-5, atol=1e-8):
        """Asynchronous form on device tensors: q_out float64 [N*S, A] (what ``greedy_actions_device`` takes with global
        states), sweeps_out int32 [N]; only enqueues."""
        _check(self._lib.mp_vi_solve_batch(self._h, model._h, float(gamma), int(iterations), float(rtol), float(atol),
                                           _ptr(q_out), _ptr(sweeps_out), MP_MEM_DEVICE))

    def load_joint(self, transitions, rewards, terminals=None, done_rule="source", available=None):
        """A joint environment of M models (agents/robust/robust.py:9-26): transitions int [M,S,A], rewards [M,S,A],
        terminals [M,S] (each model's own flags) or None.  available: bool [M,S,A], what each model's env lists in
        get_available_actions() (the joint env lists the union over its models' states, robust.py:22-25); None = all."""
        t = np.ascontiguousarray(transitions, dtype=np.int64)
        r = np.ascontiguousarray(rewards, dtype=np.float64)
        if t.shape != r.shape or t.ndim != 3:
            raise ValueError("transitions and rewards must both be [M, S, A]")
        m, s, a = t.shape
        term = None if terminals is None else np.ascontiguousarray(np.asarray(terminals).reshape(m, s).astype(np.uint8))
        h = _vp()
        _check(self._lib.mp_model_load_joint(self._h, m, s, a, _ptr(t), _ptr(r), _ptr(term), int(done_rule == "next"),
                                             C.byref(h)))
        model = Model(self, h, MODE_DETERMINISTIC, m, s, a, 0)
        if available is not None:
            av = np.ascontiguousarray(np.asarray(available).reshape(m, s, a).astype(np.uint8))
            _check(self._lib.mp_model_set_available_joint(h, _ptr(av)))
            model.available = av.astype(bool)
        return model

    def load_dense(self, transition, reward, terminal=None):
        """Dense model: transition float [S,A,S] or [M,S,A,S] (numpy -> copied; torch cuda tensor -> borrowed)."""
        on_device = hasattr(transition, "data_ptr")
        if on_device:
            t, r, term = transition, reward, terminal
            if not (t.is_contiguous() and r.is_contiguous()):
                raise ValueError("device arrays must be contiguous")
            shape = tuple(t.shape)
            keep = (t, r, term)
        else:
            t = np.ascontiguousarray(transition, dtype=np.float64)
            r = np.ascontiguousarray(reward, dtype=np.float64)
            shape = t.shape
            s_ = shape[-1]
            term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(s_).astype(np.uint8))
            keep = None
        if len(shape) not in (3, 4) or shape[-1] != shape[-3]:
            raise ValueError("transition must be [S, A, S] (or [M, S, A, S])")
        m = 1 if len(shape) == 3 else shape[0]
        s, a = shape[-3], shape[-2]
        h = _vp()
        _check(self._lib.mp_model_load_dense(self._h, m, s, a, _ptr(t), _ptr(r), _ptr(term),
                                             MP_MEM_DEVICE if on_device else MP_MEM_HOST, C.byref(h)))
        mod = Model(self, h, MODE_STOCHASTIC, m, s, a, 0)
        mod._keep = keep
        return mod

    def load_dense_rows(self, transition, reward, terminal=None):
        """Row block of a dense model: transition [S_rows,A,S] or [M,S_rows,A,S] (numpy copied / torch borrowed),
        reward [..,S_rows,A], terminal [S_rows] flags of the owned rows."""
        on_device = hasattr(transition, "data_ptr")
        if on_device:
            t, r, term = transition, reward, terminal
            shape = tuple(t.shape)
            keep = (t, r, term)
        else:
            t = np.ascontiguousarray(transition, dtype=np.float64)
            r = np.ascontiguousarray(reward, dtype=np.float64)
            shape = t.shape
            term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).astype(np.uint8))
            keep = None
        m = 1 if len(shape) == 3 else shape[0]
        rows, a, cols = shape[-3], shape[-2], shape[-1]
        h = _vp()
        _check(self._lib.mp_model_load_dense_rows(self._h, m, rows, a, cols, _ptr(t), _ptr(r), _ptr(term),
                                                  MP_MEM_DEVICE if on_device else MP_MEM_HOST, C.byref(h)))
        mod = Model(self, h, MODE_STOCHASTIC, m, rows, a, 0)
        mod.S_cols = cols
        mod._keep = keep
        return mod

    def vi_backup(self, model, gamma, v, q_out=None, robust=False):
        """One Bellman backup of a dense / row-block model. numpy in -> numpy out; torch tensors -> enqueued only."""
        if hasattr(v, "data_ptr"):
            _check(self._lib.mp_vi_backup(self._h, model._h, float(gamma), int(bool(robust)), _ptr(v), _ptr(q_out),
                                          MP_MEM_DEVICE))
            return q_out
        v = np.ascontiguousarray(v, dtype=np.float64)
        q = np.zeros((model.S, model.A), dtype=np.float64) if q_out is None else q_out
        _check(self._lib.mp_vi_backup(self._h, model._h, float(gamma), int(bool(robust)), _ptr(v), _ptr(q), MP_MEM_HOST))
        return q

    def load_sparse(self, transition, next_states, reward, terminal=None):
        t = np.ascontiguousarray(transition, dtype=np.float64)
        n = np.ascontiguousarray(next_states, dtype=np.int64)
        r = np.ascontiguousarray(reward, dtype=np.float64)
        if t.shape != n.shape or t.ndim != 3:
            raise ValueError("next and transition must both be [S, A, B]")
        s, a, b = t.shape
        term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(s).astype(np.uint8))
        h = _vp()
        _check(self._lib.mp_model_load_sparse(self._h, s, a, b, _ptr(t), _ptr(n), _ptr(r), _ptr(term), C.byref(h)))
        return Model(self, h, MODE_SPARSE, 1, s, a, b)

    def load_cartpole(self, params):
        """Closed-form CartPole model from ``CartPoleEnv.cartpole_params()`` (dict)."""
        cp = CartPoleParams(**{k: params[k] for k, _ in CartPoleParams._fields_})
        if libm_sincos_variant() not in (1, 2) and not os.environ.get("MP_CARTPOLE_SINCOS"):
            # neither restated glibc form reproduces THIS host's libm sin / cos: the device falls back to its own math library
            # and plans may differ from the reference's in the last bits of a pole angle -- say so, loudly, once per load
            import warnings
            warnings.warn("CartPole model: no restated form of sin / cos matches this host's libm (mp_libm_sincos_variant() == 0): "
                          "the device uses its own sincos and bit-exact parity with the CPU reference is NOT guaranteed",
                          RuntimeWarning, stacklevel=2)
        h = _vp()
        _check(self._lib.mp_model_load_cartpole(self._h, C.addressof(cp), C.byref(h)))
        return Model(self, h, MODE_CARTPOLE, 1, 0, 2, 0)

    # ---- value iteration ---------------------------------------------------------------------
    def vi_solve(self, model, gamma, iterations, robust=False, rtol=1e-5, atol=1e-8):
        """-> (Q [S,A] float64, sweeps run).  value_iteration.py:42-45,65-73 / robust_value_iteration.py:39-58."""
        q = np.zeros((model.S, model.A), dtype=np.float64)
        sweeps = np.zeros(1, dtype=np.int32)
        _check(self._lib.mp_vi_solve(self._h, model._h, float(gamma), int(iterations), float(rtol), float(atol),
                                     int(bool(robust)), _ptr(q), _ptr(sweeps), MP_MEM_HOST))
        if sweeps[0] < 0:       # (the host-mode call re-solves by itself when the persistent kernel gives up: not reached)
            raise NativeError(MP_ERR_HIP, "value iteration failed on the device (sweeps = {})".format(int(sweeps[0])))
        return q, int(sweeps[0])

    def vi_solve_v(self, model, gamma, iterations, rtol=1e-5, atol=1e-8, robust=False):
        """get_state_value: value_iteration.py:37-40, or robust_value_iteration.py:32-37 with robust=True."""
        v = np.zeros(model.S, dtype=np.float64)
        fn = self._lib.mp_vi_solve_v_robust if robust else self._lib.mp_vi_solve_v
        _check(fn(self._h, model._h, float(gamma), int(iterations), float(rtol), float(atol), _ptr(v), MP_MEM_HOST))
        return v

    def vi_solve_device(self, model, gamma, iterations, q_out, sweeps_out, robust=False, rtol=1e-5, atol=1e-8):
        """Asynchronous solve into device tensors (q_out float64 [S,A], sweeps_out int32 [1]).  Nothing is read back:
        ``sweeps_out[0] == -1`` (with NaN in ``q_out``) reports that the single-launch solver of small deterministic
        models could not keep its grid resident (GPU shared with other work) -- check it where the result is consumed
        (:func:`check_device_sweeps`), or use :meth:`vi_solve`, which re-solves by itself."""
        _check(self._lib.mp_vi_solve(self._h, model._h, float(gamma), int(iterations), float(rtol), float(atol),
                                     int(bool(robust)), _ptr(q_out), _ptr(sweeps_out), MP_MEM_DEVICE))

    def vi_sweeps(self, model, gamma, sweeps, robust=False):
        _check(self._lib.mp_vi_sweeps(self._h, model._h, float(gamma), int(sweeps), int(bool(robust))))

    def vi_dense_mode(self, mode):
        """How dense models are contracted with V: ``"mfma"`` (f64 matrix cores, tolerance parity) or ``"exact"``
        (numpy's own order of roundings and additions: Q, V and sweep counts bit-equal to the reference's)."""
        _check(self._lib.mp_vi_dense_mode(self._h, {"mfma": 0, "exact": 1}[mode]))

    # ---- tree search -------------------------------------------------------------------------
    def load_policy(self, model, prior, rollout, listed=None, rollout_slots=None):
        """Per-state prior / rollout policies [S, A] of a table model (mcts_with_prior.py:47-62) -> Policy.
        listed: bool [S, A], the actions the prior policy lists per state (restricted action sets, mcts.py:59-97):
        only those get a child at expansion; None = all.  rollout_slots: uint8 [S, A], the order in which the rollout
        policy lists the columns of each state when it is not the column order (mp_policy_load_ordered).
        On a batch model the tables may also be [S_each, A]: rows over the LOCAL states of one MDP that serve every MDP of the
        batch (mp_policy_load_rows)."""
        pr = np.ascontiguousarray(prior, dtype=np.float64)
        ro = pr if rollout is prior else np.ascontiguousarray(rollout, dtype=np.float64)
        rows = model.S
        if getattr(model, "n_models", 1) > 1 and pr.shape == (model.S_each, model.A):
            rows = model.S_each
        if pr.shape != (rows, model.A) or ro.shape != (rows, model.A):
            raise ValueError("prior / rollout must be [S, A] = [{}, {}]".format(model.S, model.A))
        h = _vp()
        li = None if listed is None else np.ascontiguousarray(np.asarray(listed).reshape(rows, model.A).astype(np.uint8))
        sl = None if rollout_slots is None else np.ascontiguousarray(np.asarray(rollout_slots).reshape(rows, model.A).astype(np.uint8))
        _check(self._lib.mp_policy_load_rows(self._h, model._h, _ptr(pr), _ptr(ro), _ptr(li), _ptr(sl), int(rows), C.byref(h)))
        return Policy(self, h, model)

    def uct_plan(self, model, root_state, episodes, horizon, gamma, temperature, prior_p, rollout_p, rng_state,
                 root_steps=None, max_plan_len=None, policy=None, out=None, model_index=None):
        """MCTS.plan for a batch of roots (host arrays). rng_state uint64 [n,6] is advanced in place (or a
        :class:`DeviceRng`, advanced on the device).  policy (load_policy): per-state policies instead of prior_p /
        rollout_p.  out (plan_buffers): caller-owned pinned result arrays; only the outputs they hold are produced."""
        if model.mode == MODE_CARTPOLE:      # roots are (x, x_dot, theta, theta_dot) rows
            rs = np.ascontiguousarray(root_state, dtype=np.float64).reshape(-1, 4)
        else:
            rs = np.ascontiguousarray(root_state, dtype=np.int32).reshape(-1)
        n = rs.shape[0]
        st = None if root_steps is None else np.ascontiguousarray(root_steps, dtype=np.int32).reshape(n)
        mem, rng_ptr = self._rng_arg(rng_state, n)
        mi = None
        if model_index is not None:             # batch model: root i plans on MDP model_index[i] from its LOCAL state
            mi = np.ascontiguousarray(model_index, dtype=np.int32).reshape(n)
            if policy is not None:              # (policies are tables over the global states: no second index needed)
                rs = (mi * np.int32(model.S_each) + rs).astype(np.int32)
                mi = None
        scratch = None
        if out is not None:                     # caller-owned (pinned) buffers: only the outputs they hold are produced
            mpl = out["plans"].shape[1] if "plans" in out else 0
            for k in out:
                if k != "root_state" and out[k].shape[0] != n:
                    raise ValueError("output buffer '{}' holds {} roots, the batch has {}".format(k, out[k].shape[0], n))
        elif (n <= self.SMALL_PLAN_ROOTS and mem == MP_MEM_HOST and mi is None and model.mode != MODE_CARTPOLE
              and not os.environ.get("MP_NO_PINNED_SCRATCH")):
            mpl = int(horizon if max_plan_len is None else max_plan_len)
            a_ = int(model.A)
            # (the arrays and their addresses are looked up by the plan's shape alone: building the spec and asking numpy for ten
            # ctypes addresses cost a single agent's act() ~10 us of its ~120 -- tools/act_profile.py)
            small = self.__dict__.setdefault("_small_plans", {})
            hit = small.get((n, mpl, a_))
            if hit is None or hit[0]._ptr is None:       # (evicted from the pinned-array cache: freed)
                scratch = self._scratch("uct", n, dict(
                    root_state=((n,), np.int32), root_steps=((n,), np.int32), rng=((n, 6), np.uint64), plans=((n, mpl), np.int32),
                    plan_len=((n,), np.int32), root_value=((n,), np.float64), root_child_count=((n, a_), np.int64),
                    root_child_value=((n, a_), np.float64), env_steps=((n,), np.int64)))
                if len(small) >= 8:
                    small.clear()
                hit = small[(n, mpl, a_)] = (scratch, {k: scratch[k].ctypes.data for k in (
                    "root_state", "root_steps", "rng", "plans", "plan_len", "root_value", "root_child_count", "root_child_value",
                    "env_steps")})
            scratch, sp = hit
            scratch["root_state"][:] = rs
            scratch["rng"][:] = rng_state.reshape(n, 6)
            scratch["plans"][:] = -1
            if st is not None:
                scratch["root_steps"][:] = st
            out = {k: scratch[k] for k in ("plans", "plan_len", "root_value", "root_child_count", "root_child_value", "env_steps")}
            if policy is None and mi is None:
                pp = np.ascontiguousarray(prior_p, dtype=np.float64)
                rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
                if pp.shape != (model.A,) or rp.shape != (model.A,):
                    raise ValueError("prior_p / rollout_p must have one entry per action")
                _check(self._lib.mp_uct_plan(self._h, model._h, n, sp["root_state"], None if st is None else sp["root_steps"],
                                             int(episodes), int(horizon), float(gamma), float(temperature), pp.ctypes.data,
                                             rp.ctypes.data, sp["rng"], mpl, sp["plans"], sp["plan_len"], sp["root_value"],
                                             sp["root_child_count"], sp["root_child_value"], sp["env_steps"], mem))
                return self._from_scratch(scratch, out, rng_state)
            rs, rng_ptr = scratch["root_state"], sp["rng"]
            if st is not None:
                st = scratch["root_steps"]
        else:
            mpl = int(horizon if max_plan_len is None else max_plan_len)
            out = dict(plans=np.full((n, mpl), -1, np.int32), plan_len=np.zeros(n, np.int32),
                       root_value=np.zeros(n, np.float64), root_child_count=np.zeros((n, model.A), np.int64),
                       root_child_value=np.zeros((n, model.A), np.float64), env_steps=np.zeros(n, np.int64))
        o = [_ptr(out[k]) if k in out else None for k in ("plans", "plan_len", "root_value", "root_child_count",
                                                           "root_child_value", "env_steps")]
        if policy is not None:
            _check(self._lib.mp_uct_plan_policy(self._h, model._h, policy._h, n, _ptr(rs), _ptr(st), int(episodes),
                                                int(horizon), float(gamma), float(temperature), rng_ptr, mpl,
                                                o[0], o[1], o[2], o[3], o[4], o[5], mem))
            return self._from_scratch(scratch, out, rng_state)
        pp = np.ascontiguousarray(prior_p, dtype=np.float64)
        rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
        if pp.shape != (model.A,) or rp.shape != (model.A,):
            raise ValueError("prior_p / rollout_p must have one entry per action")
        if mi is not None:
            _check(self._lib.mp_uct_plan_models(self._h, model._h, n, _ptr(mi), _ptr(rs), _ptr(st), int(episodes), int(horizon),
                                                float(gamma), float(temperature), _ptr(pp), _ptr(rp), rng_ptr, mpl,
                                                o[0], o[1], o[2], o[3], o[4], o[5], mem))
            return out
        _check(self._lib.mp_uct_plan(self._h, model._h, n, _ptr(rs), _ptr(st), int(episodes), int(horizon),
                                     float(gamma), float(temperature), _ptr(pp), _ptr(rp), rng_ptr, mpl,
                                     o[0], o[1], o[2], o[3], o[4], o[5], mem))
        return self._from_scratch(scratch, out, rng_state)

    @staticmethod
    def _from_scratch(scratch, out, rng_state):
        """Results of a small plan leave the context's pinned scratch arrays as the caller's own copies (the generator records
        go back into the caller's array, advanced)."""
        if scratch is None:
            return out
        rng_state.reshape(-1, 6)[:] = scratch["rng"]
        return {k: v.copy() for k, v in out.items()}

    @staticmethod
    def _rng_arg(rng_state, n):
        """(mem flags, pointer) for the generator records of a host-array call: a uint64 [n, 6] numpy array (copied in
        and out) or a :class:`DeviceRng` (resident: MP_MEM_RNG_DEVICE)."""
        if isinstance(rng_state, DeviceRng):
            if rng_state.n < n:
                raise ValueError("the device generator holds {} records, the batch has {} roots".format(rng_state.n, n))
            return MP_MEM_HOST | MP_MEM_RNG_DEVICE, rng_state.ptr(0)
        if not (isinstance(rng_state, np.ndarray) and rng_state.dtype == np.uint64 and rng_state.flags.c_contiguous
                and rng_state.size == n * 6):
            raise ValueError("rng_state must be a C-contiguous uint64 array of shape [n_roots, 6] or a DeviceRng")
        return MP_MEM_HOST, rng_state.ctypes.data

    def uct_plan_device(self, model, n_roots, root_state, episodes, horizon, gamma, temperature, prior_p, rollout_p,
                        rng_state, max_plan_len, plans=None, plan_len=None, root_value=None, root_child_count=None,
                        root_child_value=None, env_steps=None, root_steps=None, policy=None, model_index=None):
        """Same, on device tensors (torch); only enqueues on the ctx stream.  model_index (int32 device tensor): batch
        model, one MDP per root, root_state LOCAL."""
        if model_index is not None and policy is None:
            pp = np.ascontiguousarray(prior_p, dtype=np.float64)
            rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
            _check(self._lib.mp_uct_plan_models(self._h, model._h, int(n_roots), _ptr(model_index), _ptr(root_state),
                                                _ptr(root_steps), int(episodes), int(horizon), float(gamma), float(temperature),
                                                _ptr(pp), _ptr(rp), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                                _ptr(plan_len), _ptr(root_value), _ptr(root_child_count),
                                                _ptr(root_child_value), _ptr(env_steps), MP_MEM_DEVICE))
            return
        if model_index is not None:
            raise ValueError("per-state policies on a batch model are indexed by global state: pass global root states")
        if policy is not None:
            _check(self._lib.mp_uct_plan_policy(self._h, model._h, policy._h, int(n_roots), _ptr(root_state),
                                                _ptr(root_steps), int(episodes), int(horizon), float(gamma),
                                                float(temperature), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                                _ptr(plan_len), _ptr(root_value), _ptr(root_child_count),
                                                _ptr(root_child_value), _ptr(env_steps), MP_MEM_DEVICE))
            return
        pp = np.ascontiguousarray(prior_p, dtype=np.float64)
        rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
        _check(self._lib.mp_uct_plan(self._h, model._h, int(n_roots), _ptr(root_state), _ptr(root_steps),
                                     int(episodes), int(horizon), float(gamma), float(temperature), _ptr(pp),
                                     _ptr(rp), _ptr(rng_state), int(max_plan_len), _ptr(plans), _ptr(plan_len),
                                     _ptr(root_value), _ptr(root_child_count), _ptr(root_child_value),
                                     _ptr(env_steps), MP_MEM_DEVICE))

    def uct_plan_stochastic(self, model, root_state, episodes, horizon, gamma, temperature, prior_p, rollout_p, rng_state,
                            env_rng_state=None, closed_loop=False, root_steps=None, max_plan_len=None, policy=None, visits=None):
        """MCTS.plan on a stochastic (dense / sparse) finite-MDP model, open or closed loop (mp_uct_plan_stochastic).
        env_rng_state uint64 [n,6]: the env generator's record per root at plan time (every episode's clone starts from
        it; not advanced).  closed_loop: plans alternate action, observation key (next state index), action, ..."""
        rs = np.ascontiguousarray(root_state, dtype=np.int32).reshape(-1)
        n = rs.shape[0]
        st = None if root_steps is None else np.ascontiguousarray(root_steps, dtype=np.int32).reshape(n)
        mem, rng_ptr = self._rng_arg(rng_state, n)
        erng = None if env_rng_state is None else np.ascontiguousarray(env_rng_state, dtype=np.uint64).reshape(n, 6)
        mpl = int((2 if closed_loop else 1) * horizon if max_plan_len is None else max_plan_len)
        out = dict(plans=np.full((n, mpl), -1, np.int32), plan_len=np.zeros(n, np.int32),
                   root_value=np.zeros(n, np.float64), root_child_count=np.zeros((n, model.A), np.int64),
                   root_child_value=np.zeros((n, model.A), np.float64), env_steps=np.zeros(n, np.int64))
        if visits is not None:          # int32 [n, S]: how often the plan's env steps land in each state (mp_uct_record_visits)
            if visits.dtype != np.int32 or visits.shape != (n, model.S) or not visits.flags.c_contiguous:
                raise ValueError("visits must be a C-contiguous int32 [n_roots, S] array")
            _check(self._lib.mp_uct_record_visits(self._h, _ptr(visits)))
        if policy is not None:          # per-state policies (load_policy on this stochastic model)
            _check(self._lib.mp_uct_plan_stochastic_policy(self._h, model._h, policy._h, n, _ptr(rs), _ptr(st), int(episodes),
                                                           int(horizon), float(gamma), float(temperature), int(bool(closed_loop)),
                                                           rng_ptr, _ptr(erng), mpl, _ptr(out["plans"]), _ptr(out["plan_len"]),
                                                           _ptr(out["root_value"]), _ptr(out["root_child_count"]),
                                                           _ptr(out["root_child_value"]), _ptr(out["env_steps"]), mem))
            return out
        pp = np.ascontiguousarray(prior_p, dtype=np.float64)
        rp = np.ascontiguousarray(rollout_p, dtype=np.float64)
        if pp.shape != (model.A,) or rp.shape != (model.A,):
            raise ValueError("prior_p / rollout_p must have one entry per action")
        _check(self._lib.mp_uct_plan_stochastic(self._h, model._h, n, _ptr(rs), _ptr(st), int(episodes), int(horizon),
                                                float(gamma), float(temperature), _ptr(pp), _ptr(rp), int(bool(closed_loop)),
                                                rng_ptr, _ptr(erng), mpl, _ptr(out["plans"]), _ptr(out["plan_len"]),
                                                _ptr(out["root_value"]), _ptr(out["root_child_count"]),
                                                _ptr(out["root_child_value"]), _ptr(out["env_steps"]), mem))
        return out

    def uct_stoch_tree(self, root):
        """Tree of `root` after the last uct_plan_stochastic, creation order: parent, action (= key: action id or observed
        state), is_obs, count, value."""
        c = c_i32()
        _check(self._lib.mp_uct_stoch_tree_capacity(self._h, C.byref(c)))
        cap = c.value
        t = dict(parent=np.zeros(cap, np.int32), action=np.zeros(cap, np.int32), is_obs=np.zeros(cap, np.uint8),
                 count=np.zeros(cap, np.int64), value=np.zeros(cap, np.float64))
        n = c_i32()
        _check(self._lib.mp_uct_stoch_tree_export(self._h, int(root), cap, C.byref(n), _ptr(t["parent"]), _ptr(t["action"]),
                                                  _ptr(t["is_obs"]), _ptr(t["count"]), _ptr(t["value"])))
        t["prior"] = np.zeros(cap, np.float64)      # stored child priors (per-state policies; zeros otherwise)
        _check(self._lib.mp_uct_stoch_tree_priors(self._h, cap, _ptr(t["prior"])))
        return {k: v[:n.value].copy() for k, v in t.items()}

    def uct_step_tree(self, actions):
        """step_strategy 'subtree': keep, for the next uct_plan, the subtree under each root's child actions[i] (a host
        array, or a contiguous int32 device tensor: only enqueues then)."""
        if hasattr(actions, "data_ptr"):
            _check(self._lib.mp_uct_step_tree(self._h, int(actions.shape[0]), _ptr(actions), MP_MEM_DEVICE))
            return
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
        _check(self._lib.mp_uct_step_tree(self._h, a.shape[0], _ptr(a), MP_MEM_HOST))

    def uct_reset_tree(self):
        _check(self._lib.mp_uct_reset_tree(self._h))

    def uct_tree(self, root, cap=None):
        if cap is None:
            c = c_i32()
            _check(self._lib.mp_uct_tree_capacity(self._h, C.byref(c)))
            cap = c.value
        t = dict(parent=np.zeros(cap, np.int32), action=np.zeros(cap, np.int32), count=np.zeros(cap, np.int64),
                 value=np.zeros(cap, np.float64), first_child=np.zeros(cap, np.int32), n_children=np.zeros(cap, np.int32))
        n = c_i32()
        _check(self._lib.mp_uct_tree_export(self._h, int(root), int(cap), C.byref(n), _ptr(t["parent"]),
                                            _ptr(t["action"]), _ptr(t["count"]), _ptr(t["value"]),
                                            _ptr(t["first_child"]), _ptr(t["n_children"])))
        return {k: v[:n.value].copy() for k, v in t.items()}

    def uct_path_count(self, root, actions):
        """Visit count of the node the action sequence reaches in the last plan's tree of `root` (-1: it leaves the tree)."""
        a = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
        c = C.c_int64()
        _check(self._lib.mp_uct_path_count(self._h, int(root), _ptr(a) if a.size else None, int(a.size), C.byref(c)))
        return int(c.value)

    def opd_plan(self, model, root_state, budget, gamma, terminal_reward, rng_state, max_plan_len=64, model_index=None):
        """OptimisticDeterministicPlanner.plan for a batch of roots (host arrays).  model_index: batch model, root i plans
        on MDP model_index[i] from its LOCAL state."""
        rs = np.ascontiguousarray(root_state, dtype=np.int32).reshape(-1)
        n = rs.shape[0]
        if not (isinstance(rng_state, np.ndarray) and rng_state.dtype == np.uint64 and rng_state.flags.c_contiguous
                and rng_state.size == n * 6):
            raise ValueError("rng_state must be a C-contiguous uint64 array of shape [n_roots, 6]")
        mpl = int(max_plan_len)
        if n <= self.SMALL_PLAN_ROOTS and model_index is None and not os.environ.get("MP_NO_PINNED_SCRATCH"):
            scratch = self._scratch("opd", n, dict(
                root_state=((n,), np.int32), rng=((n, 6), np.uint64), plans=((n, mpl), np.int32), plan_len=((n,), np.int32),
                root_lower=((n,), np.float64), root_upper=((n,), np.float64), env_steps=((n,), np.int64), status=((n,), np.int32)))
            scratch["root_state"][:] = rs
            scratch["rng"][:] = rng_state.reshape(n, 6)
            scratch["plans"][:] = -1
            out = {k: scratch[k] for k in ("plans", "plan_len", "root_lower", "root_upper", "env_steps", "status")}
            _check(self._lib.mp_opd_plan(self._h, model._h, n, _ptr(scratch["root_state"]), int(budget), float(gamma),
                                         float(terminal_reward), _ptr(scratch["rng"]), mpl, _ptr(out["plans"]),
                                         _ptr(out["plan_len"]), _ptr(out["root_lower"]), _ptr(out["root_upper"]),
                                         _ptr(out["env_steps"]), _ptr(out["status"]), MP_MEM_HOST))
            return self._from_scratch(scratch, out, rng_state)
        out = dict(plans=np.full((n, mpl), -1, np.int32), plan_len=np.zeros(n, np.int32),
                   root_lower=np.zeros(n, np.float64), root_upper=np.zeros(n, np.float64),
                   env_steps=np.zeros(n, np.int64), status=np.zeros(n, np.int32))
        if model_index is not None:
            mi = np.ascontiguousarray(model_index, dtype=np.int32).reshape(n)
            _check(self._lib.mp_opd_plan_models(self._h, model._h, n, _ptr(mi), _ptr(rs), int(budget), float(gamma),
                                                float(terminal_reward), _ptr(rng_state), mpl, _ptr(out["plans"]),
                                                _ptr(out["plan_len"]), _ptr(out["root_lower"]), _ptr(out["root_upper"]),
                                                _ptr(out["env_steps"]), _ptr(out["status"]), MP_MEM_HOST))
            return out
        _check(self._lib.mp_opd_plan(self._h, model._h, n, _ptr(rs), int(budget), float(gamma),
                                     float(terminal_reward), _ptr(rng_state), mpl, _ptr(out["plans"]),
                                     _ptr(out["plan_len"]), _ptr(out["root_lower"]), _ptr(out["root_upper"]),
                                     _ptr(out["env_steps"]), _ptr(out["status"]), MP_MEM_HOST))
        return out

    def opd_plan_device(self, model, n_roots, root_state, budget, gamma, terminal_reward, rng_state, max_plan_len,
                        plans=None, plan_len=None, root_lower=None, root_upper=None, env_steps=None, status=None, model_index=None):
        if model_index is not None:
            _check(self._lib.mp_opd_plan_models(self._h, model._h, int(n_roots), _ptr(model_index), _ptr(root_state), int(budget),
                                                float(gamma), float(terminal_reward), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                                _ptr(plan_len), _ptr(root_lower), _ptr(root_upper), _ptr(env_steps),
                                                _ptr(status), MP_MEM_DEVICE))
            return
        _check(self._lib.mp_opd_plan(self._h, model._h, int(n_roots), _ptr(root_state), int(budget), float(gamma),
                                     float(terminal_reward), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                     _ptr(plan_len), _ptr(root_lower), _ptr(root_upper), _ptr(env_steps),
                                     _ptr(status), MP_MEM_DEVICE))

    def ropd_plan(self, model, root_state, budget, gamma, terminal_reward, rng_state, max_plan_len=64):
        """DiscreteRobustPlanner.plan for a batch of roots of a joint model; root_state int [n] (every model starts in
        the same state) or [n, M]."""
        rs = np.asarray(root_state, dtype=np.int32)
        if rs.ndim == 1:
            rs = np.repeat(rs[:, None], model.M, axis=1)
        rs = np.ascontiguousarray(rs.reshape(-1, model.M))
        n = rs.shape[0]
        if not (isinstance(rng_state, np.ndarray) and rng_state.dtype == np.uint64 and rng_state.flags.c_contiguous
                and rng_state.size == n * 6):
            raise ValueError("rng_state must be a C-contiguous uint64 array of shape [n_roots, 6]")
        mpl = int(max_plan_len)
        out = dict(plans=np.full((n, mpl), -1, np.int32), plan_len=np.zeros(n, np.int32),
                   root_lower=np.zeros(n, np.float64), root_upper=np.zeros(n, np.float64),
                   env_steps=np.zeros(n, np.int64), status=np.zeros(n, np.int32))
        _check(self._lib.mp_ropd_plan(self._h, model._h, n, _ptr(rs), int(budget), float(gamma), float(terminal_reward),
                                      _ptr(rng_state), mpl, _ptr(out["plans"]), _ptr(out["plan_len"]),
                                      _ptr(out["root_lower"]), _ptr(out["root_upper"]), _ptr(out["env_steps"]),
                                      _ptr(out["status"]), MP_MEM_HOST))
        return out

    def ropd_plan_device(self, model, n_roots, root_state, budget, gamma, terminal_reward, rng_state, max_plan_len,
                         plans=None, plan_len=None, root_lower=None, root_upper=None, env_steps=None, status=None):
        _check(self._lib.mp_ropd_plan(self._h, model._h, int(n_roots), _ptr(root_state), int(budget), float(gamma),
                                      float(terminal_reward), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                      _ptr(plan_len), _ptr(root_lower), _ptr(root_upper), _ptr(env_steps),
                                      _ptr(status), MP_MEM_DEVICE))

    def ropd_tree(self, root, cap, n_models):
        m = int(n_models)
        t = dict(parent=np.zeros(cap, np.int32), action=np.zeros(cap, np.int32), state=np.zeros((cap, m), np.int32),
                 depth=np.zeros(cap, np.int32), reward=np.zeros((cap, m), np.float64), lower=np.zeros((cap, m), np.float64),
                 upper=np.zeros((cap, m), np.float64), done=np.zeros((cap, m), np.uint8), count=np.zeros(cap, np.int64),
                 first_child=np.zeros(cap, np.int32), n_children=np.zeros(cap, np.int32))
        n = c_i32()
        _check(self._lib.mp_ropd_tree_export(self._h, int(root), int(cap), C.byref(n), _ptr(t["parent"]),
                                             _ptr(t["action"]), _ptr(t["state"]), _ptr(t["depth"]), _ptr(t["reward"]),
                                             _ptr(t["lower"]), _ptr(t["upper"]), _ptr(t["done"]), _ptr(t["count"]),
                                             _ptr(t["first_child"]), _ptr(t["n_children"])))
        return {k: v[:n.value].copy() for k, v in t.items()}

    def opd_tree(self, root, cap):
        t = dict(parent=np.zeros(cap, np.int32), action=np.zeros(cap, np.int32), state=np.zeros(cap, np.int32),
                 depth=np.zeros(cap, np.int32), reward=np.zeros(cap, np.float64), lower=np.zeros(cap, np.float64),
                 upper=np.zeros(cap, np.float64), done=np.zeros(cap, np.uint8), count=np.zeros(cap, np.int64),
                 first_child=np.zeros(cap, np.int32), n_children=np.zeros(cap, np.int32))
        n = c_i32()
        _check(self._lib.mp_opd_tree_export(self._h, int(root), int(cap), C.byref(n), _ptr(t["parent"]),
                                            _ptr(t["action"]), _ptr(t["state"]), _ptr(t["depth"]), _ptr(t["reward"]),
                                            _ptr(t["lower"]), _ptr(t["upper"]), _ptr(t["done"]), _ptr(t["count"]),
                                            _ptr(t["first_child"]), _ptr(t["n_children"])))
        return {k: v[:n.value].copy() for k, v in t.items()}


class PinnedArrays(dict):
    """``{name: ndarray}`` views of one pinned host allocation (mp_host_alloc); 64-byte aligned members."""

    def __init__(self, ctx, spec):
        super(PinnedArrays, self).__init__()
        self.ctx = ctx
        offs, total = {}, 0
        for name, (shape, dtype) in spec.items():
            offs[name] = total
            total += (int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize + 63) & ~63
        self._ptr = _vp()
        _check(ctx._lib.mp_host_alloc(ctx._h, max(total, 64), C.byref(self._ptr)))
        self.nbytes = total
        raw = np.ctypeslib.as_array((C.c_uint8 * max(total, 64)).from_address(self._ptr.value))
        for name, (shape, dtype) in spec.items():
            n = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
            self[name] = raw[offs[name]:offs[name] + n].view(dtype).reshape(shape)

    def close(self):
        if getattr(self, "_ptr", None) is not None and getattr(self.ctx, "_h", None):
            self.clear()
            self.ctx._lib.mp_host_free(self.ctx._h, self._ptr)
        self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceRng(object):
    """numpy-PCG64 generator records of a batch of roots, resident on the device (mp_rng): the planners advance them in
    place and nothing crosses PCIe until ``get`` is asked for."""

    def __init__(self, ctx, n):
        self.ctx, self.n = ctx, int(n)
        self._h = _vp()
        _check(ctx._lib.mp_rng_create(ctx._h, self.n, C.byref(self._h)))

    def set(self, states, first=0):
        st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 6)
        _check(self.ctx._lib.mp_rng_set(self._h, int(first), st.shape[0], _ptr(st)))

    def get(self, first=0, count=None):
        count = self.n - int(first) if count is None else int(count)
        out = np.zeros((count, 6), dtype=np.uint64)
        _check(self.ctx._lib.mp_rng_get(self._h, int(first), count, _ptr(out)))
        return out

    def seed_sequence(self, entropy, first_key=0, first=0, count=None):
        """Record first + i <- Generator(PCG64(SeedSequence(list(entropy) + [first_key + i]))) (see seed_sequence_states)."""
        count = self.n - int(first) if count is None else int(count)
        w = entropy_words(entropy) if not (hasattr(entropy, "__len__") and len(entropy) == 0) else np.zeros(0, np.uint32)
        _check(self.ctx._lib.mp_rng_seed_sequence(self._h, int(first), count, _ptr(w) if len(w) else None, len(w),
                                                  int(first_key)))

    def ptr(self, first=0):
        p = self.ctx._lib.mp_rng_device_ptr(self._h, int(first))
        if not p:
            raise NativeError(MP_ERR_ARG, load().mp_last_error().decode("utf-8", "replace"))
        return p

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.mp_rng_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model(object):
    """Device-resident transition model (mp_model)."""

    def __init__(self, ctx, handle, mode, m, s, a, b):
        self.ctx, self._h, self.mode, self.M, self.S, self.A, self.B = ctx, handle, mode, m, s, a, b
        self._keep = None
        self.n_models, self.S_each = 1, s       # batch models (Context.load_table_batch): N MDPs of S_each states

    def set_available(self, available):
        """Restrict the action sets (mp_model_set_available): bool [S, A] over this model's (global) states."""
        av = np.ascontiguousarray(np.asarray(available).reshape(self.S, self.A).astype(np.uint8))
        _check(self.ctx._lib.mp_model_set_available(self._h, _ptr(av)))
        self.available = av.astype(bool)

    def update_tables(self, first, transition, reward, terminal=None):
        """Replace the tables of MDPs [first, first + count) of a batch model -- or the whole of a single table model with
        first = 0 -- (mp_model_update_tables): transition int [count,S,A] LOCAL states, reward [count,S,A], terminal
        [count,S] iff the model has terminal flags.  Stream-ordered, no synchronisation; policies loaded for the model
        become invalid."""
        t = np.ascontiguousarray(transition, dtype=np.int64).reshape(-1, self.S_each, self.A)
        r = np.ascontiguousarray(reward, dtype=np.float64).reshape(t.shape)
        term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(t.shape[0], self.S_each).astype(np.uint8))
        _check(self.ctx._lib.mp_model_update_tables(self._h, int(first), int(t.shape[0]), _ptr(t), _ptr(r), _ptr(term)))

    def update_rows(self, rows, transition, reward, terminal=None):
        """Delta upload (mp_model_update_rows): rows int [k] GLOBAL state ids (distinct), transition int [k,A] LOCAL next
        states of each row's MDP, reward [k,A], terminal [k] or None = the rows' flags do not change."""
        rw = np.ascontiguousarray(rows, dtype=np.int32).reshape(-1)
        t = np.ascontiguousarray(transition, dtype=np.int64).reshape(rw.shape[0], self.A)
        r = np.ascontiguousarray(reward, dtype=np.float64).reshape(t.shape)
        term = None if terminal is None else np.ascontiguousarray(np.asarray(terminal).reshape(rw.shape[0]).astype(np.uint8))
        if len(np.unique(rw)) != len(rw):
            raise ValueError("update_rows: row ids must be distinct")
        _check(self.ctx._lib.mp_model_update_rows(self._h, int(rw.shape[0]), _ptr(rw), _ptr(t), _ptr(r), _ptr(term)))

    def set_episode_rules(self, done_rule="source", max_steps=0):
        """Terminal convention and TimeLimit of the env stepping this model (dense / sparse models: table models get them
        at load time)."""
        _check(self.ctx._lib.mp_model_set_episode_rules(self._h, int(done_rule == "next"), int(max_steps or 0)))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.mp_model_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StateAwarePlanners(object):
    """A batch of device-resident StateAwarePlanner objects (mp_saopd): state values, state-node lists and the node
    arena persist across plan() calls, as in the reference (tree_search/state_aware.py:76-83)."""

    def __init__(self, ctx, model, n_planners):
        self.ctx, self.model, self.n = ctx, model, int(n_planners)
        self._h = _vp()
        _check(ctx._lib.mp_saopd_create(ctx._h, model._h, self.n, C.byref(self._h)))

    def plan(self, root_state, budget, gamma, terminal_reward, rng_state, accuracy=0.0, backup_aggregated_nodes=True,
             prune_suboptimal_leaves=True, max_plan_len=None):
        rs = np.ascontiguousarray(root_state, dtype=np.int32).reshape(-1)
        if rs.shape[0] != self.n:
            raise ValueError("one root state per planner ({}), got {}".format(self.n, rs.shape[0]))
        if not (isinstance(rng_state, np.ndarray) and rng_state.dtype == np.uint64 and rng_state.flags.c_contiguous
                and rng_state.size == self.n * 6):
            raise ValueError("rng_state must be a C-contiguous uint64 array of shape [n_planners, 6]")
        mpl = int(budget + 1 if max_plan_len is None else max_plan_len)
        out = dict(plans=np.full((self.n, mpl), -1, np.int32), plan_len=np.zeros(self.n, np.int32),
                   env_steps=np.zeros(self.n, np.int64), updates=np.zeros(self.n, np.int64),
                   status=np.zeros(self.n, np.int32))
        _check(self.ctx._lib.mp_saopd_plan(self.ctx._h, self._h, _ptr(rs), int(budget), float(gamma),
                                           float(terminal_reward), float(accuracy), int(bool(backup_aggregated_nodes)),
                                           int(bool(prune_suboptimal_leaves)), _ptr(rng_state), mpl, _ptr(out["plans"]),
                                           _ptr(out["plan_len"]), _ptr(out["env_steps"]), _ptr(out["updates"]),
                                           _ptr(out["status"]), MP_MEM_HOST))
        return out

    def plan_device(self, root_state, budget, gamma, terminal_reward, rng_state, max_plan_len, accuracy=0.0,
                    backup_aggregated_nodes=True, prune_suboptimal_leaves=True, plans=None, plan_len=None, env_steps=None,
                    updates=None, status=None):
        """Asynchronous form: every array is a device buffer (torch tensors on the context's device: root_state int32
        [n], rng_state uint64-as-int64 [n, 6], plans int32 [n, max_plan_len], plan_len / status int32 [n], env_steps /
        updates int64 [n]); one launch on the context's stream, nothing read back.  A planner whose backup queue fills up
        reports MP_ERR_ALLOC in ``status`` and stays failed for every later call (mi355plan.h)."""
        _check(self.ctx._lib.mp_saopd_plan(self.ctx._h, self._h, _ptr(root_state), int(budget), float(gamma),
                                           float(terminal_reward), float(accuracy), int(bool(backup_aggregated_nodes)),
                                           int(bool(prune_suboptimal_leaves)), _ptr(rng_state), int(max_plan_len), _ptr(plans),
                                           _ptr(plan_len), _ptr(env_steps), _ptr(updates), _ptr(status), MP_MEM_DEVICE))

    def info(self):
        n, nn, root, s = c_i32(), c_i32(), c_i32(), c_i32()
        _check(self.ctx._lib.mp_saopd_info(self._h, C.byref(n), C.byref(nn), C.byref(root), C.byref(s)))
        return dict(n_planners=n.value, n_nodes=nn.value, root=root.value, n_states=s.value)

    def export(self, planner=0, current_tree_only=True):
        """Arena of one planner; current_tree_only: the nodes of the last plan's tree, ids re-based at its root."""
        inf = self.info()
        cap = inf["n_nodes"]
        t = dict(parent=np.zeros(cap, np.int32), action=np.zeros(cap, np.int32), state=np.zeros(cap, np.int32),
                 depth=np.zeros(cap, np.int32), reward=np.zeros(cap, np.float64), lower=np.zeros(cap, np.float64),
                 done=np.zeros(cap, np.uint8), count=np.zeros(cap, np.int64), first_child=np.zeros(cap, np.int32),
                 alive=np.zeros(cap, np.uint8))
        sv = np.zeros(inf["n_states"], np.float64)
        _check(self.ctx._lib.mp_saopd_export(self._h, int(planner), cap, _ptr(t["parent"]), _ptr(t["action"]),
                                             _ptr(t["state"]), _ptr(t["depth"]), _ptr(t["reward"]), _ptr(t["lower"]),
                                             _ptr(t["done"]), _ptr(t["count"]), _ptr(t["first_child"]), _ptr(t["alive"]),
                                             _ptr(sv)))
        # Rows of actions the environment does not list (restricted action sets, deterministic.py:32-35) are PHANTOMS in
        # the arena -- lower = -inf, never alive -- that only keep the ids of an expansion's |A| slots contiguous: they
        # are not nodes of the tree.  Drop them and renumber; a node's children are then contiguous from first_child,
        # n_children of them.
        lo = inf["root"] if current_tree_only else 0
        keep = ~np.isneginf(t["lower"])
        new_id = np.cumsum(keep) - 1
        a = self.model.A
        fc_old = t["first_child"].copy()
        has = fc_old >= 0
        # an expansion's |A| slots fc .. fc + |A| - 1: how many are real nodes, and the first of them
        csum = np.concatenate([[0], np.cumsum(keep)])
        fc0 = np.where(has, fc_old, 0)
        n_children = np.where(has, csum[np.minimum(fc0 + a, cap)] - csum[fc0], 0).astype(np.int32)
        next_kept = np.where(keep, np.arange(cap), cap)
        next_kept = np.minimum.accumulate(next_kept[::-1])[::-1] if cap else next_kept      # first kept row at or after i
        first_kept = np.where(n_children > 0, next_kept[fc0], -1)      # (every slot a phantom: no children, no first child)
        t["first_child"] = first_kept.astype(np.int32)
        t["n_children"] = n_children
        sel = keep.copy()
        sel[:lo] = False
        base = int(new_id[lo]) if lo < cap else 0
        out = {k: v[sel].copy() for k, v in t.items()}
        for k in ("parent", "first_child"):
            out[k] = np.where(out[k] >= lo, new_id[np.maximum(out[k], 0)] - base, -1).astype(np.int32)
        return out, sv

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.mp_saopd_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Policy(object):
    """Device-resident per-state prior / rollout policy tables (mp_policy)."""

    def __init__(self, ctx, handle, model=None):
        self.ctx, self._h, self.model = ctx, handle, model      # keeps the model it is tied to alive

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self.ctx._lib.mp_policy_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def check_device_sweeps(sweeps_out):
    """Raise if an asynchronous ``vi_solve_device`` reported failure (sweeps = -1); returns the sweep count."""
    n = int(sweeps_out.reshape(-1)[0].item())
    if n < 0:
        raise NativeError(MP_ERR_HIP, "value iteration failed on the device: the persistent kernel's grid was not "
                                      "resident (set MP_VI_NO_PERSIST=1 or use Context.vi_solve)")
    return n


def rng_state_from_generator(gen):
    """numpy Generator(PCG64) -> the 6 x uint64 state record the kernels step."""
    st = gen.bit_generator.state
    if st["bit_generator"] != "PCG64":
        raise ValueError("planner randomness must be a numpy PCG64 generator")
    s, inc = st["state"]["state"], st["state"]["inc"]
    m = (1 << 64) - 1
    return np.array([s >> 64, s & m, inc >> 64, inc & m, st["has_uint32"], st["uinteger"]], dtype=np.uint64)


def generator_set_state(gen, state6):
    """Write a stepped 6 x uint64 record back into a numpy Generator (keeps host and device streams one)."""
    st = gen.bit_generator.state
    st["state"]["state"] = (int(state6[0]) << 64) | int(state6[1])
    st["state"]["inc"] = (int(state6[2]) << 64) | int(state6[3])
    st["has_uint32"] = int(state6[4])
    st["uinteger"] = int(state6[5])
    gen.bit_generator.state = st
