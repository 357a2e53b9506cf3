"""The native (C++) create_proof of the library — `zk_plonk_create_proof`, csrc/prover.hip — behind the same Python signature as plonk.create_proof.

The reference's host side is compiled code (Rust: halo2_proofs::plonk::create_proof, called at circuits/src/sgx_dcap_verifier.rs:814-822); with no Rust
toolchain in the build image its per-proof path is written in C++ on top of the C ABI, and this module only marshals a ProvingKey (keygen.py: device
buffers and program handles) into the `zk_plonk_pk_desc` the C entry point takes.  plonk/prover.py remains as the readable twin and as the multi-GPU
(sharded tables / coset-sharded quotient) driver; both emit identical bytes for identical inputs and draws (tests/test_native_prover.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from .._lib import _dptr
from ..fields import rand_fr_array
from ..kzg import ParamsKZG
from .keygen import ProvingKey

RNG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_void_p)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class TorchExchange:
    """The collective of a sharded proof over torch.distributed (backend "nccl" = RCCL over xGMI on a multi-GPU node; "gloo" in the CPU tests, where the
    emulator's device memory is host memory): two byte tensors the library uses as its exchange buffers (zk_plonk_pk_desc.xchg_send / xchg_recv), so that
    the all-gather runs between HBM buffers with no host hop; `all_gather(send_ptr, recv_ptr, nbytes)` is what zk_allgather_fn calls."""

    def __init__(self, world: int, cap_bytes: int, device: str = "cpu", sync=None, stage_through_host: bool = False):
        """device: where the LIBRARY's device memory lives as torch sees it — "cuda" on a GPU box, "cpu" under the kernel emulator (its device memory is host
        memory).  stage_through_host: the process group cannot move device tensors (gloo with the library on a real GPU — a single-GPU dry run of the N > 1
        path): the collective then runs on host copies, the buffers the library reads and writes stay device memory."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.world, self.cap, self.device, self.sync, self.staged = torch, dist, world, int(cap_bytes), device, sync, stage_through_host
        self.send = torch.zeros(self.cap, dtype=torch.uint8, device=device)
        self.recv = torch.zeros(self.cap * world, dtype=torch.uint8, device=device)
        if device != "cpu":
            torch.cuda.synchronize()
        self.calls, self.bytes = 0, 0

    def all_gather(self, send_ptr: int, recv_ptr: int, nbytes: int) -> None:
        assert send_ptr == self.send.data_ptr() and recv_ptr == self.recv.data_ptr() and nbytes <= self.cap, "the library must use the exchange buffers it was given"
        if self.sync is not None:
            self.sync()                                             # the library's stream wrote `send` (it has synchronised already; belt and braces)
        if self.staged:
            mine = self.send[:nbytes].cpu()
            out = self.torch.empty(self.world * nbytes, dtype=self.torch.uint8)
            self.dist.all_gather_into_tensor(out, mine)
            self.recv[: self.world * nbytes].copy_(out)
        else:
            self.dist.all_gather_into_tensor(self.recv[: self.world * nbytes], self.send[:nbytes])
        if self.device != "cpu":
            self.torch.cuda.synchronize()                           # ... and will read `recv` right after the callback returns
        self.calls += 1
        self.bytes += nbytes


class PkDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("k", C.c_uint32), ("extended_k", C.c_uint32), ("cs_degree", C.c_uint32), ("blinding_factors", C.c_uint32),
                ("n_fixed", C.c_uint32), ("n_advice", C.c_uint32), ("n_instance", C.c_uint32), ("n_lookups", C.c_uint32), ("n_perm_columns", C.c_uint32),
                ("perm_columns", C.c_void_p),
                ("advice_queries", C.c_void_p), ("n_advice_queries", C.c_uint32),
                ("fixed_queries", C.c_void_p), ("n_fixed_queries", C.c_uint32),
                ("srs_g", C.c_uint64), ("srs_g_lagrange", C.c_uint64), ("program", C.c_uint64),
                ("lookup_input_programs", C.c_void_p), ("lookup_table_programs", C.c_void_p), ("lookup_table_key", C.c_void_p),
                ("fixed_values", C.c_void_p), ("fixed_polys", C.c_void_p), ("fixed_cosets", C.c_void_p),
                ("sigma_values", C.c_void_p), ("sigma_polys", C.c_void_p), ("sigma_cosets", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active_row", C.c_void_p),
                ("transcript_repr", C.c_void_p), ("transcript", C.c_uint32), ("draw_schedule", C.c_uint32),
                ("shard_world", C.c_uint32), ("shard_rank", C.c_uint32), ("allgather", ALLGATHER_FN), ("allgather_user", C.c_void_p),
                ("xchg_send", C.c_void_p), ("xchg_recv", C.c_void_p), ("xchg_cap", C.c_size_t),
                ("coset_fixed", C.c_void_p), ("coset_sigma", C.c_void_p), ("coset_l", C.c_void_p)]


class NativeProver:
    """marshalled once per (params, pk); create_proof per proof"""

    TRANSCRIPTS = {"blake2b": 0, "poseidon": 1, "evm": 2}

    def __init__(self, params: ParamsKZG, pk: ProvingKey, transcript: str = "blake2b", exchange=None):
        """exchange (sharded params only): the ranks' collective — an object with `all_gather(send_ptr, recv_ptr, nbytes)` and, optionally, the device
        buffers `send` / `recv` / `cap` the library should exchange through (TorchExchange above); every rank must prove with the same witness and rng stream."""
        assert (params.world == 1) == (pk.coset_parts is None or pk.pieces_from_cosets), "a sharded SRS goes with a proving key built on it (keygen(params.sharded(..)))"
        assert params.world == 1 or exchange is not None, "a sharded proof needs the ranks' all-gather"
        self.params, self.pk, self.be = params, pk, pk.backend
        cs = pk.vk.cs
        self._keep = []

        def u32(vals):
            a = np.ascontiguousarray(np.asarray(vals, dtype=np.int64).astype(np.uint32).reshape(-1))
            self._keep.append(a)
            return a.ctypes.data if a.size else None

        def u64(vals):
            a = np.ascontiguousarray(np.asarray(vals, dtype=np.uint64).reshape(-1))
            self._keep.append(a)
            return a.ctypes.data if a.size else None

        def ptrs(bufs):
            arr = (C.c_void_p * max(1, len(bufs)))(*[_dptr(b) for b in bufs])
            self._keep.append(arr)
            return C.cast(arr, C.c_void_p).value
        keys, key_ids = {}, []
        for lk in cs.lookups:
            key_ids.append(keys.setdefault(tuple(lk.table_expressions), len(keys)))
        repr_bytes = np.frombuffer(int(pk.vk.transcript_repr).to_bytes(32, "little"), dtype=np.uint8).copy()
        self._keep.append(repr_bytes)
        d = PkDesc()
        d.struct_size = C.sizeof(PkDesc)
        assert self.be.lib.zk_abi_struct_size(b"zk_plonk_pk_desc") == C.sizeof(PkDesc), "PkDesc has fallen behind include/zkmi355.h"
        d.k, d.extended_k, d.cs_degree, d.blinding_factors = params.k, pk.domain.extended_k, cs.degree(), cs.blinding_factors()
        d.n_fixed, d.n_advice, d.n_instance = cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns
        d.n_lookups, d.n_perm_columns = len(cs.lookups), len(cs.permutation_columns)
        d.perm_columns = u32([v for t, i in cs.permutation_columns for v in (t, i)])
        aq, fq = cs.advice_queries(), cs.fixed_queries()
        d.advice_queries, d.n_advice_queries = u32([v for c, r in aq for v in (c, r)]), len(aq)
        d.fixed_queries, d.n_fixed_queries = u32([v for c, r in fq for v in (c, r)]), len(fq)
        d.srs_g, d.srs_g_lagrange, d.program = params.g.handle, params.g_lagrange.handle, pk.evaluator.handle
        d.lookup_input_programs = u64([a.handle for a, _ in pk.lookup_compressors])
        d.lookup_table_programs = u64([b.handle for _, b in pk.lookup_compressors])
        d.lookup_table_key = u32(key_ids)
        d.fixed_values, d.fixed_polys, d.fixed_cosets = ptrs(pk.fixed_values), ptrs(pk.fixed_polys), ptrs(pk.fixed_cosets)
        d.sigma_values, d.sigma_polys, d.sigma_cosets = ptrs(pk.sigma_values), ptrs(pk.sigma_polys), ptrs(pk.sigma_cosets)
        if params.world == 1 and pk.pieces_from_cosets:
            # no extended forms: cosets 0 .. cs_degree-2 of the key's columns, the library evaluates the quotient there and calls zk_cosets_to_pieces_dev
            d.fixed_cosets = d.sigma_cosets = None
            mine = sorted(pk.coset_parts)
            assert mine == list(range(cs.degree() - 1))
            d.coset_fixed = ptrs([c for j in mine for c in pk.coset_parts[j]["fixed"]])
            d.coset_sigma = ptrs([c for j in mine for c in pk.coset_parts[j]["sigma"]])
            d.coset_l = ptrs([c for j in mine for c in pk.coset_parts[j]["l"]])
        elif params.world == 1:
            d.l0, d.l_last, d.l_active_row = _dptr(pk.l0), _dptr(pk.l_last), _dptr(pk.l_active_row)
        else:
            # one proof over the ranks: this rank's table slices are behind the SRS handles already (ParamsKZG.sharded); the key's cosets come per coset
            # (keygen on sharded params keeps only this rank's: pk.coset_parts), and the collective is the caller's
            mine = sorted(pk.coset_parts)
            d.shard_world, d.shard_rank = params.world, params.rank
            d.coset_fixed = ptrs([c for j in mine for c in pk.coset_parts[j]["fixed"]])
            d.coset_sigma = ptrs([c for j in mine for c in pk.coset_parts[j]["sigma"]])
            d.coset_l = ptrs([c for j in mine for c in pk.coset_parts[j]["l"]])
            self.exchange, self.comm_errors = exchange, []

            def gather(_user, send, recv, nbytes):
                try:
                    exchange.all_gather(int(send), int(recv), int(nbytes))
                    return 0
                except BaseException as e:                          # never let an exception cross the FFI: the proof ends with ZK_ERR_COMM
                    self.comm_errors.append(e)
                    return 1
            self._gather_cb = ALLGATHER_FN(gather)
            d.allgather = self._gather_cb
            if getattr(exchange, "send", None) is not None:
                d.xchg_send, d.xchg_recv, d.xchg_cap = exchange.send.data_ptr(), exchange.recv.data_ptr(), exchange.cap
        d.transcript_repr = repr_bytes.ctypes.data
        d.transcript = self.TRANSCRIPTS[transcript]         # which Fiat-Shamir transcript / proof encoding (zk_plonk_pk_desc.transcript)
        d.draw_schedule = 1                                 # halo2_proofs v2023_01_20's order of Fr::random draws (prover.py draw_plan): the only schedule the library knows
        self.desc = d
        self.proof_cap = (64 if transcript == "evm" else 32) * (cs.num_advice_columns + 3 * len(cs.lookups) + len(cs.permutation_columns) + 16 +
                               len(aq) + len(fq) + 1 + len(cs.permutation_columns) + 3 * len(cs.permutation_columns) + 5 * len(cs.lookups) + 8)

    def create_proof(self, advice: Sequence, instances: Sequence[Sequence[int]], rng) -> bytes:
        """advice: host (n, 4) uint64 arrays (page-locked or ordinary) or device buffers — all of one kind; instances: canonical ints per instance column;
        rng: numpy Generator (seeded) or fields.OsRng().  Returns the proof bytes."""
        be, d = self.be, self.desc
        on_device = not isinstance(advice[0], np.ndarray) if len(advice) else False
        keep = [np.ascontiguousarray(a, dtype=np.uint64) for a in advice] if not on_device else []
        adv = (C.c_void_p * max(1, len(advice)))(*([a.ctypes.data for a in keep] if not on_device else [_dptr(a) for a in advice]))
        inst = [np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in col) or bytes(32), dtype=np.uint8).copy() for col in instances]
        inst_ptrs = (C.c_void_p * max(1, len(inst)))(*[a.ctypes.data for a in inst])
        lens = (C.c_uint32 * max(1, len(inst)))(*[len(col) for col in instances])
        errors = []

        def draw(_user, count, out):                                   # the caller's Fr::random: same sampler, same order as the Python twin
            try:
                a = rand_fr_array(rng, int(count))
                C.memmove(out, a.ctypes.data, a.nbytes)
            except BaseException as e:                                  # never let an exception cross the FFI
                errors.append(e)
        cb = RNG_FN(draw)
        out = np.empty(self.proof_cap, dtype=np.uint8)
        ln = C.c_size_t()
        rc = be.lib.zk_plonk_create_proof(be.ctx, C.byref(d), adv, C.c_int(1 if on_device else 0), inst_ptrs, lens, cb, None,
                                          out.ctypes.data_as(C.c_void_p), C.c_size_t(out.size), C.byref(ln))
        if errors:
            raise errors[0]
        if getattr(self, "comm_errors", None):
            raise self.comm_errors.pop()
        be._ck(rc)
        ph = (C.c_double * 9)()
        be.lib.zk_plonk_last_phase_ms(ph)
        self.phase_ms = dict(zip(("1_instances", "2_advice_commit", "3_lookup_permuted", "4_grand_products", "5_random_poly", "6_ntt_evaluate_h", "7_h_construct_commit",
                                  "8_evaluations", "9_shplonk"), [round(v, 2) for v in ph]))
        return out[: ln.value].tobytes()


def create_proof_native(params: ParamsKZG, pk: ProvingKey, advice, instances, rng) -> bytes:
    return NativeProver(params, pk).create_proof(advice, instances, rng)
