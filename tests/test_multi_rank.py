"""N>1 path on CPU: two gloo ranks, base table sharded by index range, partial points all-gathered and
summed on every rank (what bench.py --gpus N does over RCCL).  Checked against the oracle's full MSM."""
import os
import socket
import tempfile

import numpy as np
import torch.multiprocessing as mp

import multi_rank_worker


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_msm_two_ranks_gloo(built):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(multi_rank_worker.run, args=(2, _free_port(), 150, d), nprocs=2, join=True)
        for r in range(2):
            assert np.load(os.path.join(d, f"rank{r}.npy"))[0] == 1


import pytest


@pytest.mark.parametrize("world", [4, 8])
def test_sharded_create_proof_more_ranks_than_two(built, world):
    """4 ranks: one coset of the quotient each (the toy circuit's extended domain has 4); 8 ranks: more ranks than cosets — the two ranks of a coset
    each evaluate half of its rows (zk_quotient_run_coset_rows_dev), the all-gather carries 8 half-cosets — and every rank emits the golden proof."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(multi_rank_worker.run_sharded_proof, args=(world, _free_port(), d), nprocs=world, join=True)
        for r in range(world):
            assert np.load(os.path.join(d, f"proof_rank{r}.npy"))[0] == 1


def test_sharded_create_proof_two_ranks_gloo(built):
    """SURVEY §8e / BASELINE configs[4]: the prover with BOTH SRS tables sharded by index range over two ranks — every commitment of the proof is
    two partial MSMs + one all_gather — emits byte for byte the golden proof of the single-device prover, on every rank."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(multi_rank_worker.run_sharded_proof, args=(2, _free_port(), d), nprocs=2, join=True)
        for r in range(2):
            assert np.load(os.path.join(d, f"proof_rank{r}.npy"))[0] == 1
