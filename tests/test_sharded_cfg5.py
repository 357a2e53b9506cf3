"""BASELINE configs[4] / SURVEY 8d cfg 5 and 8e at THEIR OWN SIZES on the one GPU a lease has: the sharded paths, not only the workload.
  (a) the 2^21 MSM over 8 ranks: 8 contexts, each with 1/8 of the base table resident (index-range shards, 16 MiB of points = 256 MiB expanded per rank), its slice
      of the scalars, zk_msm_partial_dev -> 128-byte XYZZ partial -> zk_g1_sum_xyzz on the gathered eight; closed form [sum s_i k_i] G.
  (b) the x4 census proof (100 advice columns, 44 lookups, 20 permutation sets: 269 commitments) at k = 21, single context AND sharded over 2 ranks: sharded SRS tables and
      sharded keys, 8 all-gathers per proof (7 commitment phases + the numerators, here barrier + device copies between two contexts), both ranks' bytes equal to the
      single-context native proof of the same witness and draws — which verify_proof accepts.
What an 8-GPU node adds to these is xGMI under the collective (libzkmi355_rccl.so), nothing else.  Reference: roadmap only, /root/reference README.md:23-47."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest

import zk_dcap_verifier_amd as z
from zk_dcap_verifier_amd import plonk

import parity_cases as pc
import test_create_proof as tcp

pytestmark = pytest.mark.gpu


def test_sharded_msm_2p21_eight_ranks_gpu(gpu, orc, pyref):
    import bench
    log_n, world, seed = 21, 8, 20241021
    n = 1 << log_n
    n_loc = n // world
    ks = pc.rand_fr(orc, pyref, n, seed)
    sc = pc.rand_fr(orc, pyref, n, seed + 1)
    dk, dpts = gpu.to_device(ks), gpu.alloc(n * 64)
    gpu.g1_fixed_base_mul(dk, n, dpts)                                # P_i = [k_i] G: the whole table once, on rank 0's context
    dk.upload(sc)
    ranks = [gpu] + [z.Backend(0) for _ in range(world - 1)]          # 8 contexts on the one device = 8 ranks
    handles = [be.bases_register((dpts.ptr + r * n_loc * 64, n_loc)) for r, be in enumerate(ranks)]      # rank r expands ITS index range only
    dpts.free()
    parts = np.zeros((world, 16), dtype=np.uint64)
    errs = []

    def run(r):
        try:
            parts[r] = ranks[r].msm_partial(handles[r], dk.ptr + r * n_loc * 32, n_loc)
        except BaseException as e:                                     # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    got = gpu.g1_sum_xyzz(parts)                                      # what every rank does with the all-gathered 8 x 128 bytes
    rinv = pow(1 << 256, -1, pyref.R)
    want = orc.g1_to_affine(orc.g1_mul(orc.g1_generator(), orc.ints_to_limbs([bench.dot_mod_r(ks, sc) * rinv % pyref.R])[0]))[0]
    assert (got[:8] == want).all() and got[8:].any()
    for be, h in zip(ranks, handles):
        be.bases_release(h)
    for be in ranks[1:]:
        be.close()
    dk.free()
    gpu.trim_pool()


class _Fabric:
    """the ranks' collective inside one process: a barrier and world x world device copies (tests/csrc/capi_prove.c does the same in C)"""

    def __init__(self, world):
        self.world, self.bar = world, threading.Barrier(world, timeout=600)
        self.parts, self.send = [None] * world, [0] * world

    def host_all_gather(self, rank):
        def gather(part):
            self.parts[rank] = np.ascontiguousarray(part)
            self.bar.wait()
            out = np.stack(self.parts)
            self.bar.wait()
            return out
        return gather

    def exchange(self, rank, be):
        fab = self

        class X:
            send = recv = None
            calls = 0

            def all_gather(self, send_ptr, recv_ptr, nbytes):
                fab.send[rank] = send_ptr
                fab.bar.wait()
                for r in range(fab.world):
                    be._ck(be.lib.zk_dev_copy(be.ctx, C.c_void_p(recv_ptr + r * nbytes), C.c_void_p(fab.send[r]), C.c_size_t(nbytes)))
                fab.bar.wait()
                self.calls += 1
        return X()


def test_x4_census_proof_sharded_over_two_ranks_equals_the_single_context_proof_gpu(gpu, orc):
    import verifier
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sgx_shaped_circuit as sc
    k, world = int(os.environ.get("ZK_CFG5_K", "21")), 2
    cs, fixed, asm, advice = sc.build(z, gpu, k, census="full_chain_x4", table_bits=16)
    assert (cs.num_advice_columns, len(cs.lookups), cs.degree()) == (100, 44, 5)
    full = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=gpu)
    pk = plonk.keygen(full, cs, fixed, asm)
    single = plonk.NativeProver(full, pk).create_proof(advice, [], np.random.default_rng(3))
    assert verifier.verify_proof(pk.vk, tcp.TAU, [], single) is True
    pk.release()
    g_host, gl_host = full.g_host, full.g_lagrange_host
    full.release()
    gpu.trim_pool()
    host_advice = [a.download((1 << k, 4)) if not isinstance(a, np.ndarray) else a for a in advice]
    for a in advice:
        if not isinstance(a, np.ndarray):
            a.free()
    fab = _Fabric(world)
    ranks = [gpu, z.Backend(0)]
    out, errs = [None] * world, []

    def run(r):
        try:
            be = ranks[r]
            params = z.kzg.ParamsKZG.sharded(k, g_host, gl_host, r, world, fab.host_all_gather(r), backend=be)
            key = plonk.keygen(params, cs, fixed, asm)                 # sharded key: this rank's cosets of the fixed / sigma / l columns only (device columns of one GPU are every context's)
            xch = fab.exchange(r, be)
            out[r] = plonk.NativeProver(params, key, exchange=xch).create_proof(host_advice, [], np.random.default_rng(3))
            assert xch.calls == 8, xch.calls                           # 7 commitment phases + the numerators
            key.release()
            params.release()
        except BaseException as e:                                     # noqa: BLE001
            errs.append((r, repr(e)))
            fab.bar.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert out[0] == single and out[1] == single, "a rank's sharded proof differs from the single-context proof"
    ranks[1].close()
    gpu.trim_pool()
