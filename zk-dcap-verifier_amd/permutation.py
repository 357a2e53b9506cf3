"""halo2_proofs::plonk::{permutation, lookup}::prover grand products, MI355X edition (SURVEY.md §8f "next 1").

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75) src/plonk/permutation/prover.rs `Argument::commit` and
src/plonk/lookup/prover.rs `Permuted::commit_product`, reached from the reference through create_proof
(circuits/src/sgx_dcap_verifier.rs:814-822).  Columns stay in HBM: the returned z buffers go straight into
`commit_lagrange` (zk_msm) and `lagrange_to_coeff`."""
from __future__ import annotations

import numpy as np

from ._lib import Backend, default_backend

DELTA = 0x09226B6E22C6F0CA64EC26AAD4C86E715B5F898E5E963F25870E56BBE533E9A2   # Fr::DELTA = 7^(2^28)
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _mont(x: int) -> np.ndarray:
    v = (x << 256) % R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def permutation_commit(columns, sigmas, k: int, cs_degree: int, beta, gamma, blinding_rows, backend: Backend | None = None):
    """All column sets of the permutation argument.

    columns / sigmas: device buffers (Lagrange basis) in cs.permutation.columns order; blinding_rows[s]: the
    `blinding_factors` random rows of set s (the caller's RNG draws, so proofs stay reproducible under a seeded RNG).
    Returns the list of z device buffers, one per set (chunks of cs_degree - 2 columns)."""
    be = backend or default_backend()
    chunk = cs_degree - 2
    n = 1 << k
    n_sets = (len(columns) + chunk - 1) // chunk
    zs = [be.alloc(n * 32) for _ in range(n_sets)]
    if zs:   # every set in one device call: fractions per set, ONE batch inversion, scans, host-side chaining of the set boundaries
        be.permutation_product_all_dev(columns, sigmas, chunk, k, beta, gamma, np.stack([np.asarray(b, dtype=np.uint64).reshape(-1, 4) for b in blinding_rows[:n_sets]]), zs)
    return zs


def lookup_commit_product(compressed_input, compressed_table, permuted_input, permuted_table, k: int, beta, gamma, blinding_rows,
                          backend: Backend | None = None):
    be = backend or default_backend()
    z = be.alloc((1 << k) * 32)
    be.lookup_product_dev(compressed_input, compressed_table, permuted_input, permuted_table, k, beta, gamma, blinding_rows, z)
    return z


def permute_expression_pair(compressed_input, compressed_table, k: int, blinding_factors: int, blind_input, blind_table, backend: Backend | None = None):
    """lookup::Argument::commit_permuted's permute_expression_pair on device columns -> (permuted_input, permuted_table) buffers.
    Raises ZkError if an input value is not in the table (halo2: Error::ConstraintSystemFailure)."""
    be = backend or default_backend()
    n = 1 << k
    a, s = be.alloc(n * 32), be.alloc(n * 32)
    be.lookup_permute_dev(compressed_input, compressed_table, k, blinding_factors, blind_input, blind_table, a, s)
    return a, s


def permute_expression_pairs(compressed_inputs, compressed_tables, k: int, blinding_factors: int, blind_inputs, blind_tables, backend: Backend | None = None):
    """All lookup arguments of a proof at once (the loop over `lookups` in create_proof's phase 3): one launch sequence for every sort.
    blind_inputs / blind_tables: (count, blinding_factors + 1, 4).  Returns [(permuted_input, permuted_table)] device buffers."""
    be = backend or default_backend()
    n = 1 << k
    outs = [(be.alloc(n * 32), be.alloc(n * 32)) for _ in compressed_inputs]
    if outs:
        be.lookup_permute_batch_dev(compressed_inputs, compressed_tables, k, blinding_factors, blind_inputs, blind_tables,
                                    [o[0] for o in outs], [o[1] for o in outs])
    return outs


def lookup_commit_products(quads, k: int, beta, gamma, blinding_rows, backend: Backend | None = None):
    """commit_product of every lookup of a proof in one device call; quads = [(compressed_input, compressed_table, permuted_input, permuted_table)],
    blinding_rows: (count, blinding_factors, 4).  Returns the z device buffers."""
    be = backend or default_backend()
    zs = [be.alloc((1 << k) * 32) for _ in quads]
    if zs:
        be.lookup_product_batch_dev(quads, k, beta, gamma, blinding_rows, zs)
    return zs
