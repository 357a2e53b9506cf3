"""Worker for tests/test_multi_rank.py: the N>1 MSM path of bench.py (base table sharded by index range,
128-byte XYZZ partials all-gathered, every rank sums) on the gloo backend with the kernel emulator."""
import os
import sys

import numpy as np


def run(rank: int, world: int, port: int, n: int, out_dir: str):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import oracle as orc
    import pyref
    import parity_cases as pc
    import zk_dcap_verifier_amd as z
    from conftest import EMU_SO

    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = z.Backend(0, lib_path=EMU_SO)
    be.tune(msm_sort_threads=32, msm_sort_wgs=2, msm_block=32, msm_target_threads=64, msm_min_chunk=2)
    sc, bases = pc.msm_inputs(orc, pyref, n, 1234)           # same seed on every rank = the global problem
    lo, hi = rank * n // world, (rank + 1) * n // world       # this rank's shard of bases and scalars
    h = be.bases_register(np.ascontiguousarray(bases[lo:hi]))
    ds = be.to_device(np.ascontiguousarray(sc[lo:hi]))
    part = be.msm_partial(h, ds, hi - lo)
    mine = torch.from_numpy(part.view(np.int64).copy())
    gathered = [torch.zeros(16, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, mine)
    total = be.g1_sum_xyzz(torch.stack(gathered).numpy().view(np.uint64))
    want = orc.g1_to_affine(orc.best_multiexp(sc, bases))[0]
    ok = bool((total[:8] == want).all())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()
    be.close()


def run_sharded_proof(rank: int, world: int, port: int, out_dir: str):
    """create_proof with the SRS tables sharded over `world` gloo ranks (every commitment = partial MSMs + all_gather of 128-byte points):
    every rank must emit the golden proof bytes of the single-GPU prover."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import zk_dcap_verifier_amd as z
    from zk_dcap_verifier_amd import plonk
    from zk_dcap_verifier_amd.transcript import Blake2bWrite
    from conftest import EMU_SO
    import test_create_proof as tcp

    dist.init_process_group("gloo", rank=rank, world_size=world)
    be = z.Backend(0, lib_path=EMU_SO)
    be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4, msm_target_threads=64, msm_min_chunk=2,
            vec_block=32, quot_threads=32)

    def all_gather(part):
        mine = torch.from_numpy(np.ascontiguousarray(part).view(np.int64).copy())
        out = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(out, mine)
        return torch.stack(out).numpy().view(np.uint64)
    k = 6
    full = z.kzg.ParamsKZG.setup(k, tcp.TAU, backend=be)
    params = z.kzg.ParamsKZG.sharded(k, full.g_host, full.g_lagrange_host, rank, world, all_gather, backend=be)
    full.release()
    cs, fixed, asm, advice, instances = tcp.toy_circuit(k)
    pk = plonk.keygen(params, cs, fixed, asm)
    n_cosets = 1 << (pk.domain.extended_k - k)
    assert sorted(pk.coset_parts) == params.my_cosets(n_cosets)       # the quotient is sharded by coset
    units = params.my_units(n_cosets)
    if world >= n_cosets and world % n_cosets == 0:                    # e.g. 8 ranks, 4 cosets: every rank owns one (coset, half of its rows) — nobody idles
        assert len(units) == 1 and units[0][2] == (1 << k) * n_cosets // world, units
    tr = Blake2bWrite()
    plonk.create_proof(params, pk, advice, instances, np.random.default_rng(7), tr)
    ok = tr.finalize() == tcp._golden()

    # the same with the numerators exchanged in "device" memory the caller owns (torch tensors; under the emulator device memory is host memory)
    held = {}

    def coset_exchange(nbytes):
        if held.get("n") != nbytes:
            held.update(n=nbytes, send=torch.zeros(nbytes // 8, dtype=torch.int64), recv=torch.zeros(world * nbytes // 8, dtype=torch.int64))

        def run():
            be.sync()
            dist.all_gather_into_tensor(held["recv"], held["send"])
        return held["send"].data_ptr(), held["recv"].data_ptr(), run
    params.coset_exchange = coset_exchange
    tr = Blake2bWrite()
    plonk.create_proof(params, pk, advice, instances, np.random.default_rng(7), tr)
    ok = ok and tr.finalize() == tcp._golden() and "send" in held

    # ... and through the NATIVE prover (zk_plonk_create_proof in shard mode, csrc/prover.hip): partial MSMs + one all-gather of 128-byte points per commitment
    # phase, coset / half-coset quotient + one all-gather of the numerators, all through the caller's collective on "device" buffers (zk_allgather_fn)
    params.coset_exchange = None
    xch = plonk.native.TorchExchange(world, 1 << 16, "cpu", sync=be.sync)
    native = plonk.NativeProver(params, pk, exchange=xch)
    proofs = [native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(7)) for _ in range(2)]
    n_phases = 7 + 1                                                   # advice, permuted pairs, grand products, random poly, h pieces, SHPLONK h and quotient; + the numerators
    ok = ok and proofs == [tcp._golden()] * 2 and xch.calls == 2 * n_phases
    # a failing collective ends the proof with ZK_ERR_COMM on every rank, and the next proof is unaffected
    class Broken:
        send = recv = None

        def all_gather(self, *a):
            raise RuntimeError("link down")
    try:
        plonk.NativeProver(params, pk, exchange=Broken()).create_proof([a.copy() for a in advice], instances, np.random.default_rng(7))
        ok = False
    except RuntimeError as e:
        ok = ok and "link down" in str(e)
    ok = ok and native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(7)) == tcp._golden()
    # a failure of ONE rank's own (the last rank's witness leaves a lookup table: ZK_ERR_ARG from its lookup phase, after the first exchange) must not leave the
    # others waiting in the next all-gather: the failing rank signals it in that exchange, every other rank returns ZK_ERR_COMM from it, and all of them go on
    bad_advice = tcp.toy_circuit(k, tamper="lookup")[3]
    calls_before = xch.calls
    try:
        native.create_proof([a.copy() for a in (bad_advice if rank == world - 1 else advice)], instances, np.random.default_rng(7))
        ok = False
    except z.ZkError as e:
        ok = ok and e.code == (-1 if rank == world - 1 else -6) and ("signalled to the other ranks" if rank == world - 1 else "reported a failure of its own") in str(e)
    ok = ok and xch.calls - calls_before == 2                          # the advice commitments, then the exchange that carried the mark
    ok = ok and native.create_proof([a.copy() for a in advice], instances, np.random.default_rng(7)) == tcp._golden()
    np.save(os.path.join(out_dir, f"proof_rank{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()
    be.close()
