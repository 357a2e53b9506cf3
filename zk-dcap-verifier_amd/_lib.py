"""ctypes binding of libzkmi355.so (include/zkmi355.h).  Thin: pointers and sizes only.

The product path FAILS LOUDLY when the HIP library is missing or no GPU is usable — there is no
CPU implementation behind this module.  (Tests that exercise kernel index logic on a machine
without a GPU pass an explicit ``lib_path`` to the emulator build under tests/csrc/; nothing in
this package knows about or defaults to it.)
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libzkmi355.so")

ZK_OK = 0
_ERR_NAMES = {-1: "ZK_ERR_ARG", -2: "ZK_ERR_HIP", -3: "ZK_ERR_NODEV", -4: "ZK_ERR_PROGRAM", -5: "ZK_ERR_LIMIT", -6: "ZK_ERR_COMM"}


class ZkError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


ABI_VERSION = 4          # ZK_ABI_VERSION of include/zkmi355.h this binding follows


class QuotientArgs(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("fixed", C.c_void_p), ("advice", C.c_void_p), ("instance", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("perm_cosets", C.c_void_p), ("perm_products", C.c_void_p), ("n_sets", C.c_uint32),
                ("lookup_product", C.c_void_p), ("lookup_input", C.c_void_p), ("lookup_table", C.c_void_p),
                ("challenges", C.c_void_p), ("beta", C.c_void_p), ("gamma", C.c_void_p), ("theta", C.c_void_p),
                ("y", C.c_void_p), ("out", C.c_void_p)]


def _load(path: str):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.zk_last_error.restype = C.c_char_p
    lib.zk_version.restype = C.c_char_p
    lib.zk_timing_get.restype = C.c_double
    lib.zk_abi_version.restype = C.c_uint32
    lib.zk_abi_struct_size.restype = C.c_uint32
    # the binding's own structs against the library it loaded (include/zkmi355.h, ABI versioning): a stale .so or a stale binding stops here
    if lib.zk_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: ABI version {lib.zk_abi_version()}, this binding is written against {ABI_VERSION} (rebuild: __graft_entry__.build())")
    if lib.zk_abi_struct_size(b"zk_quotient_args") != C.sizeof(QuotientArgs):
        raise RuntimeError(f"{path}: sizeof(zk_quotient_args) = {lib.zk_abi_struct_size(b'zk_quotient_args')}, the binding's QuotientArgs has {C.sizeof(QuotientArgs)}")
    return lib


def _dptr(x) -> int:
    """Device pointer of a DeviceBuffer / torch tensor / int."""
    if isinstance(x, int):
        return x
    if hasattr(x, "ptr"):
        return int(x.ptr)
    if hasattr(x, "data_ptr"):
        return int(x.data_ptr())
    raise TypeError(f"not a device buffer: {type(x)}")


class DeviceBuffer:
    """hipMalloc'ed buffer owned through the C ABI (so callers need neither torch nor HIP)."""

    def __init__(self, backend: "Backend", nbytes: int):
        self.backend, self.nbytes = backend, int(nbytes)
        with backend._pool_lock:                      # several host threads may share one Backend
            cached = backend._pool.get(self.nbytes)
            if cached:                                # hipMalloc / hipFree of 16-64 MiB columns cost far more than the kernels between them
                self.ptr = cached.pop()
                backend._pool_bytes -= self.nbytes
                return
        p = C.c_void_p()
        backend._ck(backend.lib.zk_dev_alloc(backend.ctx, C.c_size_t(self.nbytes), C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray, offset: int = 0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        b = self.backend
        b._ck(b.lib.zk_dev_upload(b.ctx, C.c_void_p(self.ptr + offset), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        return self

    def zero(self, offset: int = 0, nbytes: int | None = None):
        """zk_dev_zero: Fr zeros written on the device (no host buffer crosses the link)"""
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        assert offset + nbytes <= self.nbytes
        b = self.backend
        b._ck(b.lib.zk_dev_zero(b.ctx, C.c_void_p(self.ptr + offset), C.c_size_t(nbytes)))
        return self

    def download(self, shape, dtype=np.uint64, offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        b = self.backend
        b._ck(b.lib.zk_dev_download(b.ctx, out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), C.c_size_t(out.nbytes)))
        return out

    def copy_from(self, src, nbytes: int | None = None):
        """device -> device copy of the first nbytes of `src` into this buffer."""
        b = self.backend
        nb = self.nbytes if nbytes is None else int(nbytes)
        assert nb <= self.nbytes
        b._ck(b.lib.zk_dev_copy(b.ctx, C.c_void_p(self.ptr), C.c_void_p(_dptr(src)), C.c_size_t(nb)))
        return self

    def free(self):
        if self.ptr:
            b = self.backend
            ptr, self.ptr = self.ptr, 0
            with b._pool_lock:
                keep = bool(b.ctx) and b._pool_bytes + self.nbytes <= b.pool_limit_bytes
                if keep:
                    b._pool.setdefault(self.nbytes, []).append(ptr)       # size-keyed free list, released by Backend.close()
                    b._pool_bytes += self.nbytes
            if not keep and b.ctx:
                b.lib.zk_dev_free(b.ctx, C.c_void_p(ptr))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Backend:
    """One zk_ctx = one GPU.  Mirrors nothing in halo2 (halo2 has no device object); the
    halo2-named functions in arithmetic/domain/kzg/evaluation take a Backend (or use the default)."""

    def __init__(self, device: int = 0, lib_path: str | None = None):
        self.lib = _load(lib_path or LIB_PATH)
        self.ctx = C.c_void_p()
        rc = self.lib.zk_ctx_create(C.c_int(device), C.byref(self.ctx))
        if rc != ZK_OK:
            raise ZkError(rc, f"zk_ctx_create(device={device}) failed — no usable gfx950 GPU? (no CPU fallback)")
        self.device = device
        self._bases_cache = {}
        self._pool, self._pool_bytes, self._pool_lock = {}, 0, threading.Lock()
        self._pinned = {}
        self.pool_limit_bytes = int(os.environ.get("ZK_POOL_LIMIT_GIB", "24")) << 30   # freed device buffers kept for reuse (per context)

    # -- plumbing -------------------------------------------------------------------------------
    def _ck(self, rc: int):
        if rc != ZK_OK:
            raise ZkError(rc, (self.lib.zk_last_error(self.ctx) or b"").decode())

    def version(self) -> str:
        return self.lib.zk_version().decode()

    def trim_pool(self):
        with self._pool_lock:
            pool, self._pool, self._pool_bytes = self._pool, {}, 0
        for ptrs in pool.values():
            for p in ptrs:
                self.lib.zk_dev_free(self.ctx, C.c_void_p(p))
        self.lib.zk_plonk_trim(self.ctx)              # ... and the buffers zk_plonk_create_proof keeps per context between proofs (a k = 21 proof leaves ~80 GB of them)

    def close(self):
        if self.ctx:
            self.trim_pool()
            self.lib.zk_plonk_trim(self.ctx)
            for p in list(self._pinned.values()):
                self.lib.zk_host_free(self.ctx, C.c_void_p(p))
            self._pinned = {}
            self.lib.zk_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def tune(self, **kw):
        for k, v in kw.items():
            self._ck(self.lib.zk_tune_set(self.ctx, k.encode(), C.c_int(int(v))))

    def tune_get(self, key: str) -> int:
        v = C.c_int()
        self._ck(self.lib.zk_tune_get(self.ctx, key.encode(), C.byref(v)))
        return v.value

    def timing(self, on: bool = True):
        self._ck(self.lib.zk_timing_enable(self.ctx, C.c_int(1 if on else 0)))

    def timing_get(self, label: str):
        """(total ms, launches) accumulated since timing(True)."""
        ms = self.lib.zk_timing_get(self.ctx, label.encode())
        n = self.lib.zk_timing_get(self.ctx, (label + "#n").encode())
        return (ms, int(n)) if ms >= 0 else (None, 0)

    def stat_get(self, label: str) -> float:
        v = self.lib.zk_timing_get(self.ctx, label.encode())
        return v if v >= 0 else 0.0

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        return DeviceBuffer(self, max(arr.nbytes, 32)).upload(arr)

    def sync(self):
        self._ck(self.lib.zk_dev_sync(self.ctx))

    def host_alloc(self, shape, dtype=np.uint64) -> np.ndarray:
        """page-locked host array (zk_host_alloc) — staging memory for columns that cross PCIe every proof; released by host_free / close()"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._ck(self.lib.zk_host_alloc(self.ctx, C.c_size_t(nbytes), C.byref(p)))
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p is not None and self.ctx:
            self._ck(self.lib.zk_host_free(self.ctx, C.c_void_p(p)))

    def upload_columns(self, dev_cols, host_cols, nbytes_each: int):
        """host_cols[i] (contiguous arrays of nbytes_each) -> dev_cols[i], one call"""
        assert len(dev_cols) == len(host_cols)
        hosts = [np.ascontiguousarray(h) for h in host_cols]
        assert all(h.nbytes >= nbytes_each for h in hosts)
        harr = (C.c_void_p * max(1, len(hosts)))(*[h.ctypes.data for h in hosts])
        self._ck(self.lib.zk_dev_upload_batch(self.ctx, self._ptr_array(dev_cols), harr, C.c_size_t(len(hosts)), C.c_size_t(nbytes_each)))

    # -- MSM ------------------------------------------------------------------------------------
    def bases_register(self, bases) -> int:
        h = C.c_uint64()
        if isinstance(bases, np.ndarray):
            b = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
            self._ck(self.lib.zk_bases_register(self.ctx, b.ctypes.data_as(C.c_void_p), C.c_size_t(b.shape[0]), C.byref(h)))
        else:
            ptr, n = bases
            self._ck(self.lib.zk_bases_register_dev(self.ctx, C.c_void_p(_dptr(ptr)), C.c_size_t(n), C.byref(h)))
        return h.value

    def bases_share(self, owner: "Backend", owner_handle: int) -> int:
        h = C.c_uint64()
        self._ck(self.lib.zk_bases_share(self.ctx, owner.ctx, C.c_uint64(owner_handle), C.byref(h)))
        return h.value

    def bases_enable_runs(self, handle: int):
        """build the prefix-sum twin of a registered table (zk_bases_enable_runs): run-heavy columns are then committed through their adjacent differences"""
        self._ck(self.lib.zk_bases_enable_runs(self.ctx, C.c_uint64(handle)))

    def bases_release(self, handle: int):
        self._ck(self.lib.zk_bases_release(self.ctx, C.c_uint64(handle)))

    def msm(self, handle: int, scalars, n: int | None = None) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        if isinstance(scalars, np.ndarray):
            s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
            n = s.shape[0] if n is None else n
            self._ck(self.lib.zk_msm(self.ctx, C.c_uint64(handle), s.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        else:
            assert n is not None
            self._ck(self.lib.zk_msm_dev(self.ctx, C.c_uint64(handle), C.c_void_p(_dptr(scalars)), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def msm_batch(self, handle: int, columns, n: int | None = None) -> np.ndarray:
        """columns: list of host (n, 4) arrays or of device buffers -> (count, 12) normalised G1."""
        count = len(columns)
        out = np.zeros((count, 12), dtype=np.uint64)
        if count == 0:
            return out
        if isinstance(columns[0], np.ndarray):
            cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in columns]
            n = cols[0].shape[0] if n is None else n
            assert all(c.shape[0] >= n for c in cols)
            arr = (C.c_void_p * count)(*[c.ctypes.data for c in cols])
            self._ck(self.lib.zk_msm_batch(self.ctx, C.c_uint64(handle), arr, C.c_size_t(count), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        else:
            assert n is not None
            arr = (C.c_void_p * count)(*[_dptr(c) for c in columns])
            self._ck(self.lib.zk_msm_batch_dev(self.ctx, C.c_uint64(handle), arr, C.c_size_t(count), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def msm_partial(self, handle: int, scalars_dev, n: int) -> np.ndarray:
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.lib.zk_msm_partial_dev(self.ctx, C.c_uint64(handle), C.c_void_p(_dptr(scalars_dev)), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def msm_batch_partial(self, handle: int, columns, n: int) -> np.ndarray:
        """unnormalised XYZZ partial results (count, 16) of `count` device columns against a (sharded) table"""
        count = len(columns)
        out = np.zeros((count, 16), dtype=np.uint64)
        if count:
            arr = (C.c_void_p * count)(*[_dptr(c) for c in columns])
            self._ck(self.lib.zk_msm_batch_partial_dev(self.ctx, C.c_uint64(handle), arr, C.c_size_t(count), C.c_size_t(n), out.ctypes.data_as(C.c_void_p)))
        return out

    def g1_sum_xyzz_batch(self, parts: np.ndarray) -> np.ndarray:
        """parts: (n_parts, count, 16) partial sums as gathered from the ranks -> (count, 12) normalised points"""
        parts = np.ascontiguousarray(parts, dtype=np.uint64)
        n_parts, count = parts.shape[0], parts.shape[1]
        out = np.zeros((count, 12), dtype=np.uint64)
        rc = self.lib.zk_g1_sum_xyzz_batch(parts.ctypes.data_as(C.c_void_p), C.c_size_t(n_parts), C.c_size_t(count), out.ctypes.data_as(C.c_void_p))
        if rc:
            raise ZkError(rc, "zk_g1_sum_xyzz_batch")
        return out

    def g1_sum_xyzz(self, parts: np.ndarray) -> np.ndarray:
        parts = np.ascontiguousarray(parts, dtype=np.uint64).reshape(-1, 16)
        out = np.zeros(12, dtype=np.uint64)
        rc = self.lib.zk_g1_sum_xyzz(parts.ctypes.data_as(C.c_void_p), C.c_size_t(parts.shape[0]), out.ctypes.data_as(C.c_void_p))
        if rc:
            raise ZkError(rc, "zk_g1_sum_xyzz")
        return out

    def g1_fixed_base_mul(self, scalars_dev, n: int, out_dev):
        self._ck(self.lib.zk_g1_fixed_base_mul_dev(self.ctx, C.c_void_p(_dptr(scalars_dev)), C.c_size_t(n), C.c_void_p(_dptr(out_dev))))

    def g1_decompress_dev(self, bytes_dev, n: int, sign_bit: int, out_affine_dev) -> None:
        bad = C.c_uint32()
        self._ck(self.lib.zk_g1_decompress_dev(self.ctx, C.c_void_p(_dptr(bytes_dev)), C.c_size_t(n), C.c_uint32(sign_bit), C.c_void_p(_dptr(out_affine_dev)), C.byref(bad)))

    def g1_compress_dev(self, affine_dev, n: int, sign_bit: int, bytes_dev) -> None:
        self._ck(self.lib.zk_g1_compress_dev(self.ctx, C.c_void_p(_dptr(affine_dev)), C.c_size_t(n), C.c_uint32(sign_bit), C.c_void_p(_dptr(bytes_dev))))

    def g1_ntt_dev(self, in_dev, log_n: int, omega, scale, out_dev):
        w = self._fe(omega)
        sc = self._fe(scale) if scale is not None else None
        self._ck(self.lib.zk_g1_ntt_dev(self.ctx, C.c_void_p(_dptr(in_dev)), C.c_uint32(log_n), w.ctypes.data_as(C.c_void_p),
                                        sc.ctypes.data_as(C.c_void_p) if sc is not None else None, C.c_void_p(_dptr(out_dev))))

    # -- NTT / domain ---------------------------------------------------------------------------
    @staticmethod
    def _fe(x) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(x, dtype=np.uint64).reshape(4))

    def ntt(self, a: np.ndarray, log_n: int, omega) -> None:
        """in place on a host array (n, 4)"""
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and a.size == 4 << log_n
        w = self._fe(omega)
        self._ck(self.lib.zk_ntt(self.ctx, a.ctypes.data_as(C.c_void_p), C.c_uint32(log_n), w.ctypes.data_as(C.c_void_p)))

    def ntt_dev(self, a_dev, log_n: int, omega) -> None:
        w = self._fe(omega)
        self._ck(self.lib.zk_ntt_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_uint32(log_n), w.ctypes.data_as(C.c_void_p)))

    def lagrange_to_coeff_dev(self, a_dev, k): self._ck(self.lib.zk_lagrange_to_coeff_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_uint32(k)))
    def coeff_to_lagrange_dev(self, a_dev, k): self._ck(self.lib.zk_coeff_to_lagrange_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_uint32(k)))

    def _ptr_array(self, bufs):
        return (C.c_void_p * max(1, len(bufs)))(*[_dptr(b) for b in bufs])

    def ntt_batch_dev(self, cols, log_n: int, omega):
        w = self._fe(omega)
        self._ck(self.lib.zk_ntt_batch_dev(self.ctx, self._ptr_array(cols), C.c_size_t(len(cols)), C.c_uint32(log_n), w.ctypes.data_as(C.c_void_p)))

    def lagrange_to_coeff_batch_dev(self, cols, k):
        self._ck(self.lib.zk_lagrange_to_coeff_batch_dev(self.ctx, self._ptr_array(cols), C.c_size_t(len(cols)), C.c_uint32(k)))

    def coeff_to_extended_batch_dev(self, coeffs, outs, k, ek):
        assert len(coeffs) == len(outs)
        self._ck(self.lib.zk_coeff_to_extended_batch_dev(self.ctx, self._ptr_array(coeffs), self._ptr_array(outs), C.c_size_t(len(outs)), C.c_uint32(k), C.c_uint32(ek)))

    def coeff_to_coset_batch_dev(self, coeffs, outs, k, ek, coset: int):
        assert len(coeffs) == len(outs)
        self._ck(self.lib.zk_coeff_to_coset_batch_dev(self.ctx, self._ptr_array(coeffs), self._ptr_array(outs), C.c_size_t(len(outs)), C.c_uint32(k), C.c_uint32(ek),
                                                      C.c_uint32(coset)))

    def fr_interleave_dev(self, cosets, n: int, out_dev):
        self._ck(self.lib.zk_fr_interleave_dev(self.ctx, self._ptr_array(cosets), C.c_size_t(len(cosets)), C.c_size_t(n), C.c_void_p(_dptr(out_dev))))

    def cosets_to_pieces_dev(self, numer, k, ek, out):
        """numer[j] = the numerator's n values on coset j (clobbered), j < len(numer) = cs_degree - 1; out[i] receives piece i of h(X) (zk_cosets_to_pieces_dev)"""
        assert len(numer) == len(out)
        self._ck(self.lib.zk_cosets_to_pieces_dev(self.ctx, self._ptr_array(numer), C.c_uint32(len(numer)), C.c_uint32(k), C.c_uint32(ek), self._ptr_array(out)))

    def coeff_to_extended_dev(self, coeff_dev, k, ek, out_dev):
        self._ck(self.lib.zk_coeff_to_extended_dev(self.ctx, C.c_void_p(_dptr(coeff_dev)), C.c_uint32(k), C.c_uint32(ek), C.c_void_p(_dptr(out_dev))))

    def extended_to_coeff_dev(self, a_dev, k, ek):
        self._ck(self.lib.zk_extended_to_coeff_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_uint32(k), C.c_uint32(ek)))

    def divide_by_vanishing_poly_dev(self, a_dev, k, ek):
        self._ck(self.lib.zk_divide_by_vanishing_poly_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_uint32(k), C.c_uint32(ek)))

    # -- vectors --------------------------------------------------------------------------------
    def _vec(self, fn, a, b, out, n):
        self._ck(getattr(self.lib, fn)(self.ctx, C.c_void_p(_dptr(a)), C.c_void_p(_dptr(b)), C.c_void_p(_dptr(out)), C.c_size_t(n)))

    def fr_mul_dev(self, a, b, out, n): self._vec("zk_fr_mul_dev", a, b, out, n)
    def fr_add_dev(self, a, b, out, n): self._vec("zk_fr_add_dev", a, b, out, n)
    def fr_sub_dev(self, a, b, out, n): self._vec("zk_fr_sub_dev", a, b, out, n)
    def fq_mul_dev(self, a, b, out, n): self._vec("zk_fq_mul_dev", a, b, out, n)

    def fr_scale_dev(self, a, scalar, out, n):
        s = self._fe(scalar)
        self._ck(self.lib.zk_fr_scale_dev(self.ctx, C.c_void_p(_dptr(a)), s.ctypes.data_as(C.c_void_p), C.c_void_p(_dptr(out)), C.c_size_t(n)))

    # -- grand products ------------------------------------------------------------------------
    def permutation_product_dev(self, values, sigmas, k, beta, gamma, delta_start, z_init, blinding, z_dev) -> np.ndarray:
        bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(-1, 4))
        sc = [self._fe(v) for v in (beta, gamma, delta_start, z_init)]
        last = np.zeros(4, dtype=np.uint64)
        self._ck(self.lib.zk_permutation_product_dev(self.ctx, self._ptr_array(values), self._ptr_array(sigmas), C.c_size_t(len(values)), C.c_uint32(k),
                                                     sc[0].ctypes.data_as(C.c_void_p), sc[1].ctypes.data_as(C.c_void_p), sc[2].ctypes.data_as(C.c_void_p),
                                                     sc[3].ctypes.data_as(C.c_void_p), bl.ctypes.data_as(C.c_void_p), C.c_uint32(bl.shape[0]),
                                                     C.c_void_p(_dptr(z_dev)), last.ctypes.data_as(C.c_void_p)))
        return last

    def lookup_product_dev(self, cin, ctab, pin, ptab, k, beta, gamma, blinding, z_dev):
        bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(-1, 4))
        sc = [self._fe(v) for v in (beta, gamma)]
        self._ck(self.lib.zk_lookup_product_dev(self.ctx, C.c_void_p(_dptr(cin)), C.c_void_p(_dptr(ctab)), C.c_void_p(_dptr(pin)), C.c_void_p(_dptr(ptab)),
                                                C.c_uint32(k), sc[0].ctypes.data_as(C.c_void_p), sc[1].ctypes.data_as(C.c_void_p),
                                                bl.ctypes.data_as(C.c_void_p), C.c_uint32(bl.shape[0]), C.c_void_p(_dptr(z_dev))))

    def permutation_product_all_dev(self, values, sigmas, chunk_len: int, k: int, beta, gamma, blinding, z_devs):
        n_sets = len(z_devs)
        bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(n_sets, -1, 4))
        sc = [self._fe(v) for v in (beta, gamma)]
        self._ck(self.lib.zk_permutation_product_all_dev(self.ctx, self._ptr_array(values), self._ptr_array(sigmas), C.c_size_t(len(values)), C.c_uint32(chunk_len),
                                                         C.c_uint32(k), sc[0].ctypes.data_as(C.c_void_p), sc[1].ctypes.data_as(C.c_void_p),
                                                         bl.ctypes.data_as(C.c_void_p), C.c_uint32(bl.shape[1]), self._ptr_array(z_devs)))

    def lookup_product_batch_dev(self, quads, k, beta, gamma, blinding, z_devs):
        """quads: [(compressed_input, compressed_table, permuted_input, permuted_table)] device columns; blinding: (count, bf, 4)."""
        cnt = len(quads)
        bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(cnt, -1, 4))
        sc = [self._fe(v) for v in (beta, gamma)]
        flat = [c for q in quads for c in q]
        self._ck(self.lib.zk_lookup_product_batch_dev(self.ctx, self._ptr_array(flat), C.c_size_t(cnt), C.c_uint32(k), sc[0].ctypes.data_as(C.c_void_p),
                                                      sc[1].ctypes.data_as(C.c_void_p), bl.ctypes.data_as(C.c_void_p), C.c_uint32(bl.shape[1]),
                                                      self._ptr_array(z_devs)))

    def lookup_permute_dev(self, inp, table, k: int, blinding_factors: int, blind_input, blind_table, out_input, out_table):
        bi = np.ascontiguousarray(np.asarray(blind_input, dtype=np.uint64).reshape(blinding_factors + 1, 4))
        bt = np.ascontiguousarray(np.asarray(blind_table, dtype=np.uint64).reshape(blinding_factors + 1, 4))
        self._ck(self.lib.zk_lookup_permute_dev(self.ctx, C.c_void_p(_dptr(inp)), C.c_void_p(_dptr(table)), C.c_uint32(k), C.c_uint32(blinding_factors),
                                                bi.ctypes.data_as(C.c_void_p), bt.ctypes.data_as(C.c_void_p), C.c_void_p(_dptr(out_input)),
                                                C.c_void_p(_dptr(out_table))))

    def lookup_permute_batch_dev(self, inputs, tables, k: int, blinding_factors: int, blind_inputs, blind_tables, out_inputs, out_tables):
        cnt = len(inputs)
        bi = np.ascontiguousarray(np.asarray(blind_inputs, dtype=np.uint64).reshape(cnt, blinding_factors + 1, 4))
        bt = np.ascontiguousarray(np.asarray(blind_tables, dtype=np.uint64).reshape(cnt, blinding_factors + 1, 4))
        self._ck(self.lib.zk_lookup_permute_batch_dev(self.ctx, self._ptr_array(inputs), self._ptr_array(tables), C.c_size_t(cnt), C.c_uint32(k),
                                                      C.c_uint32(blinding_factors), bi.ctypes.data_as(C.c_void_p), bt.ctypes.data_as(C.c_void_p),
                                                      self._ptr_array(out_inputs), self._ptr_array(out_tables)))

    # -- evaluation phase -----------------------------------------------------------------------
    def eval_polynomial_batch_dev(self, polys, n: int, points) -> np.ndarray:
        pts = np.ascontiguousarray(np.asarray(points, dtype=np.uint64).reshape(-1, 4))
        assert pts.shape[0] == len(polys)
        out = np.zeros((len(polys), 4), dtype=np.uint64)
        self._ck(self.lib.zk_eval_polynomial_batch_dev(self.ctx, self._ptr_array(polys), C.c_size_t(len(polys)), C.c_size_t(n),
                                                       pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def kate_division_dev(self, a_dev, n: int, b, q_dev):
        bb = self._fe(b)
        self._ck(self.lib.zk_kate_division_dev(self.ctx, C.c_void_p(_dptr(a_dev)), C.c_size_t(n), bb.ctypes.data_as(C.c_void_p), C.c_void_p(_dptr(q_dev))))

    def fr_lincomb_dev(self, polys, scalars, n: int, out_dev):
        """out[i] = sum_j scalars[j] * polys[j][i] (SHPLONK combinations) in one pass; scalars: (count, 4) Montgomery limbs."""
        sc = np.ascontiguousarray(np.asarray(scalars, dtype=np.uint64).reshape(-1, 4))
        assert sc.shape[0] == len(polys)
        self._ck(self.lib.zk_fr_lincomb_dev(self.ctx, self._ptr_array(polys), sc.ctypes.data_as(C.c_void_p), C.c_size_t(len(polys)), C.c_size_t(n),
                                            C.c_void_p(_dptr(out_dev))))

    # -- quotient -------------------------------------------------------------------------------
    def quotient_program_load(self, blob: bytes) -> int:
        h = C.c_uint64()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        self._ck(self.lib.zk_quotient_program_load(self.ctx, buf, C.c_size_t(len(blob)), C.byref(h)))
        return h.value

    def quotient_program_share(self, owner: "Backend", owner_prog: int) -> int:
        """a handle of this context onto a program `owner` (same GPU) loaded: one compiled program per process (zk_quotient_program_share)"""
        h = C.c_uint64()
        self._ck(self.lib.zk_quotient_program_share(self.ctx, owner.ctx, C.c_uint64(owner_prog), C.byref(h)))
        return h.value

    def quotient_program_info(self, prog: int) -> dict:
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.lib.zk_quotient_program_info(self.ctx, C.c_uint64(prog), C.byref(a), C.byref(b), C.byref(c)))
        return {"instructions": a.value, "slots": b.value, "columns": c.value}

    def quotient_program_kernels(self, prog: int) -> int:
        """generated kernels the program runs (tune quot_jit at load time); 0 = the micro-op interpreter"""
        n = C.c_uint32()
        self._ck(self.lib.zk_quotient_program_kernels(self.ctx, C.c_uint64(prog), C.byref(n)))
        return n.value

    def quotient_program_opmix(self, prog: int, part: int = 0) -> dict:
        """opcode census of the compiled program; part 1 / 2 = its high / low part when it has a degree split (quotient_program_split)"""
        c = (C.c_uint32 * 9)()
        self._ck(self.lib.zk_quotient_program_part_opmix(self.ctx, C.c_uint64(prog), C.c_uint32(part), c) if part else self.lib.zk_quotient_program_opmix(self.ctx, C.c_uint64(prog), c))
        return dict(zip(("add", "sub", "mul", "sqr", "dbl", "neg", "mov", "muladd", "memory_operands"), [int(x) for x in c]))

    def quotient_program_split(self, prog: int) -> dict:
        """the degree split of a compiled program: cosets the low part runs on (0 = no split) and the instruction counts of the two parts"""
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.lib.zk_quotient_program_split(self.ctx, C.c_uint64(prog), C.byref(a), C.byref(b), C.byref(c)))
        return {"low_cosets": a.value, "instructions_high": b.value, "instructions_low": c.value}

    def quotient_program_release(self, prog: int):
        self._ck(self.lib.zk_quotient_program_release(self.ctx, C.c_uint64(prog)))

    def quotient_run_dev(self, prog: int, *, fixed, advice, instance, l0, l_last, l_active_row, perm_cosets, perm_products,
                         lookup_product, lookup_input, lookup_table, challenges, beta, gamma, theta, y, out, coset: int | None = None,
                         rows: tuple | None = None, part: int = 0, low_cosets: int = 0):
        """part 1 / 2: the high / low part of a program with a degree split (zk_quotient_run_high_dev / _low_dev / _coset_part_dev); low_cosets: the low part on the rows of
        that many cosets of extended-layout columns, `out` coset-major"""
        keep = []

        def parr(cols):
            arr = (C.c_void_p * max(1, len(cols)))(*[_dptr(c) for c in cols])
            keep.append(arr)
            return C.cast(arr, C.c_void_p)

        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64).reshape(-1, 4)) if len(challenges) else np.zeros((1, 4), np.uint64)
        sc = [self._fe(v) for v in (beta, gamma, theta, y)]
        a = QuotientArgs(C.sizeof(QuotientArgs), parr(fixed), parr(advice), parr(instance), _dptr(l0), _dptr(l_last), _dptr(l_active_row),
                         parr(perm_cosets), parr(perm_products), len(perm_products),
                         parr(lookup_product), parr(lookup_input), parr(lookup_table),
                         ch.ctypes.data, sc[0].ctypes.data, sc[1].ctypes.data, sc[2].ctypes.data, sc[3].ctypes.data, _dptr(out))
        if part:
            assert rows is None
            if coset is not None:
                self._ck(self.lib.zk_quotient_run_coset_part_dev(self.ctx, C.c_uint64(prog), C.byref(a), C.c_uint32(coset), C.c_uint32(part)))
            elif part == 1:
                self._ck(self.lib.zk_quotient_run_high_dev(self.ctx, C.c_uint64(prog), C.byref(a)))
            else:
                self._ck(self.lib.zk_quotient_run_low_dev(self.ctx, C.c_uint64(prog), C.byref(a), C.c_uint32(low_cosets)))
        elif coset is None:
            self._ck(self.lib.zk_quotient_run_dev(self.ctx, C.c_uint64(prog), C.byref(a)))
        elif rows is None:
            self._ck(self.lib.zk_quotient_run_coset_dev(self.ctx, C.c_uint64(prog), C.byref(a), C.c_uint32(coset)))
        else:
            self._ck(self.lib.zk_quotient_run_coset_rows_dev(self.ctx, C.c_uint64(prog), C.byref(a), C.c_uint32(coset), C.c_uint64(rows[0]), C.c_uint64(rows[1])))


def _host_ptrs(cols):
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    return (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols]), cols


def _pk_methods():
    def pk_load(self, prog: int, fixed, sigma, l0, l_last, l_active_row, form: int = 0) -> int:
        fa, k1 = _host_ptrs(fixed)
        sa, k2 = _host_ptrs(sigma)
        ls = [np.ascontiguousarray(c, dtype=np.uint64) for c in (l0, l_last, l_active_row)]
        h = C.c_uint64()
        self._ck(self.lib.zk_pk_load(self.ctx, C.c_uint64(prog), fa, sa, ls[0].ctypes.data_as(C.c_void_p), ls[1].ctypes.data_as(C.c_void_p),
                                     ls[2].ctypes.data_as(C.c_void_p), C.c_int(form), C.byref(h)))
        return h.value

    def pk_release(self, pk: int):
        self._ck(self.lib.zk_pk_release(self.ctx, C.c_uint64(pk)))

    def evaluate_h(self, pk: int, *, advice, instance, perm_products, lookup_product, lookup_input, lookup_table, challenges, beta, gamma, theta, y,
                   out_rows: int, finish: bool):
        arrs = [_host_ptrs(g) for g in (advice, instance, perm_products, lookup_product, lookup_input, lookup_table)]
        ch = np.ascontiguousarray(np.asarray(challenges, dtype=np.uint64).reshape(-1, 4)) if len(challenges) else np.zeros((1, 4), np.uint64)
        sc = [self._fe(v) for v in (beta, gamma, theta, y)]
        out = np.empty((out_rows, 4), dtype=np.uint64)
        self._ck(self.lib.zk_evaluate_h(self.ctx, C.c_uint64(pk), *[a[0] for a in arrs], ch.ctypes.data_as(C.c_void_p),
                                        *[v.ctypes.data_as(C.c_void_p) for v in sc], C.c_int(1 if finish else 0), out.ctypes.data_as(C.c_void_p)))
        return out
    Backend.pk_load, Backend.pk_release, Backend.evaluate_h = pk_load, pk_release, evaluate_h


_pk_methods()

_default = None
_default_lock = threading.Lock()


def default_backend() -> Backend:
    """Process-wide backend on the GPU named by LOCAL_RANK (one process per GPU), else device 0."""
    global _default
    with _default_lock:
        if _default is None:
            _default = Backend(int(os.environ.get("LOCAL_RANK", "0")))
        return _default
