"""The units a multi-GPU prover shards the quotient by: the 2^(extended_k - k) cosets of the extended domain.
zk_coeff_to_coset_batch_dev must give exactly every 2^e-th entry of coeff_to_extended, zk_fr_interleave_dev must put them back."""
import numpy as np
import pytest

import parity_cases as pc


def _check(be, orc, pyref, k, e, seed):
    n, ek = 1 << k, k + e
    polys = [pc.rand_fr(orc, pyref, n, seed + i) for i in range(3)]
    d = [be.to_device(p) for p in polys]
    ext = [be.alloc((n << e) * 32) for _ in polys]
    be.coeff_to_extended_batch_dev(d, ext, k, ek)
    full = [x.download((n << e, 4)) for x in ext]
    cos = [[be.alloc(n * 32) for _ in polys] for _ in range(1 << e)]
    for j in range(1 << e):
        be.coeff_to_coset_batch_dev(d, cos[j], k, ek, j)
        for c, f in zip(cos[j], full):
            assert (c.download((n, 4)) == f[j::1 << e]).all(), (j,)
    out = be.alloc((n << e) * 32)
    be.fr_interleave_dev([cos[j][1] for j in range(1 << e)], n, out)
    assert (out.download((n << e, 4)) == full[1]).all()


@pytest.mark.parametrize("k,e", [(4, 1), (5, 2), (6, 3)])
def test_emulated_cosets(emu, orc, pyref, k, e):
    _check(emu, orc, pyref, k, e, seed=k)


@pytest.mark.gpu
@pytest.mark.parametrize("k,e", [(10, 2), (16, 2), (19, 2), (14, 3)])
def test_gpu_cosets(gpu, orc, pyref, k, e):
    _check(gpu, orc, pyref, k, e, seed=k)


def test_coset_arguments_are_checked(emu, orc, pyref):
    import zk_dcap_verifier_amd as z
    d = emu.to_device(pc.rand_fr(orc, pyref, 16, 1))
    o = emu.alloc(16 * 32)
    with pytest.raises(z.ZkError):
        emu.coeff_to_coset_batch_dev([d], [o], 4, 6, 4)              # only cosets 0..3 exist
    with pytest.raises(z.ZkError):
        emu.coeff_to_coset_batch_dev([d], [o], 4, 3, 0)              # extended_k < k
