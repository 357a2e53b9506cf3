"""The sizes BASELINE.json names, RESULT-checked on the GPU (SURVEY.md §8d cfg 3/4):
   MSM 2^20 (seed 20241008) and 2^24 (seed 20241010): uniform scalars in [0, r), closed form [sum_i s_i k_i] G against the oracle's scalar multiplication;
   batched NTT 2^22 x 25 columns (seed 20241011) through zk_ntt_batch_dev: 64 outputs against the direct sum sum_i a_i omega^(i j) (oracle Horner at
   omega^j) and the forward/inverse round trip of every column.
The oracle cannot run a 2^24 MSM or a 2^22 NTT batch in seconds; these are the size-independent properties the domain offers."""
import numpy as np
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


def _closed_form_msm(gpu, orc, pyref, log_n, seed):
    import bench
    n = 1 << log_n
    ks = pc.rand_fr(orc, pyref, n, seed)                 # discrete logs of the bases (Montgomery limbs of uniform field elements)
    sc = pc.rand_fr(orc, pyref, n, seed + 1)             # the scalars
    assert (sc[:, 3] >> np.uint64(61)).any(), "scalars must reach above 2^253 (uniform in [0, r))"
    dk, dpts = gpu.to_device(ks), gpu.alloc(n * 64)
    gpu.g1_fixed_base_mul(dk, n, dpts)
    h = gpu.bases_register((dpts, n))
    dpts.free()
    dk.upload(sc)
    got = gpu.msm(h, dk, n)
    # values are limbs / R: sum (K_i / R)(S_i / R) = (sum K_i S_i) / R^2; as a Montgomery-form scalar for the oracle: (sum K_i S_i) / R
    rinv = pow(1 << 256, -1, pyref.R)
    total_mont = bench.dot_mod_r(ks, sc) * rinv % pyref.R
    want = orc.g1_to_affine(orc.g1_mul(orc.g1_generator(), orc.ints_to_limbs([total_mont])[0]))[0]
    assert (got[:8] == want).all() and got[8:].any()
    # the exact dot product itself is cross-checked against Python integers on a slice
    m = 1 << 12
    assert bench.dot_mod_r(ks[:m], sc[:m]) == sum(a * b for a, b in zip(orc.limbs_to_ints(ks[:m]), orc.limbs_to_ints(sc[:m]))) % pyref.R
    gpu.bases_release(h)
    dk.free()


def test_msm_closed_form_2p20_seed_20241008(gpu, orc, pyref):
    _closed_form_msm(gpu, orc, pyref, 20, 20241008)


def test_msm_closed_form_2p24_seed_20241010(gpu, orc, pyref):
    _closed_form_msm(gpu, orc, pyref, 24, 20241010)
    gpu.trim_pool()


def test_ntt_batch_2p22_x25_direct_sum_spot_check_and_round_trip(gpu, orc, pyref):
    log_n, cols = 22, 25
    n = 1 << log_n
    w = pyref.omega(log_n)
    wm = orc.fr_from_ints([w])[0]
    host = [pc.rand_fr(orc, pyref, n, 20241011 + c) for c in range(cols)]
    dev = [gpu.to_device(a) for a in host]
    gpu.ntt_batch_dev(dev, log_n, wm)
    rng = np.random.default_rng(20241011)
    picks = [(int(rng.integers(0, cols)), int(rng.integers(0, n))) for _ in range(60)] + [(0, 0), (cols - 1, n - 1), (3, n // 2), (7, 1)]
    for c, j in picks:                                   # out[j] = sum_i a[i] omega^(i j): Horner evaluation of the column at omega^j on the oracle
        got = dev[c].download((1, 4), offset=j * 32)[0]
        want = orc.eval_polynomial(host[c], orc.fr_from_ints([pow(w, j, pyref.R)])[0])
        assert (got == want).all(), (c, j)
    gpu.ntt_batch_dev(dev, log_n, orc.fr_from_ints([pow(w, -1, pyref.R)])[0])
    ninv = orc.fr_from_ints([pow(n, -1, pyref.R)])[0]
    for c in range(cols):                                # iNTT(NTT(a)) / n = a, every element of every column
        gpu.fr_scale_dev(dev[c], ninv, dev[c], n)
        assert (dev[c].download((n, 4)) == host[c]).all(), c
    for d in dev:
        d.free()
    gpu.trim_pool()
