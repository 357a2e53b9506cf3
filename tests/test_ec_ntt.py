"""NTT over G1 (g_to_lagrange) and the KZG consistency it implies: commit_lagrange(evals) == commit(coeffs)."""
import numpy as np
import pytest

import parity_cases as pc
import zk_dcap_verifier_amd as z


def _check_vs_oracle(be, orc, pyref, log_n, seed):
    n = 1 << log_n
    pts = orc.gen_bases_arith(seed + 3, seed + 11, n)
    if n > 2:
        pts[1] = 0                                        # identity among the inputs
    w = orc.fr_from_ints([pow(pyref.omega(log_n), -1, pyref.R)])[0]
    sc = orc.fr_from_ints([pow(n, -1, pyref.R)])[0]
    want = orc.g1_fft(pts, log_n, w, sc)
    d, o = be.to_device(pts), be.alloc(n * 64)
    be.g1_ntt_dev(d, log_n, w, sc, o)
    assert (o.download((n, 8)) == want).all()
    be.g1_ntt_dev(d, log_n, w, None, o)
    assert (o.download((n, 8)) == orc.g1_fft(pts, log_n, w, None)).all()
    d.free()
    o.free()


@pytest.mark.parametrize("log_n", [0, 1, 3, 5])
def test_emulated_g1_ntt(emu, orc, pyref, log_n):
    _check_vs_oracle(emu, orc, pyref, log_n, seed=log_n)


def _kzg_consistency(be, orc, pyref, k, seed):
    """With g[i] = [tau^i]G and g_lagrange = EC-iFFT(g): MSM(g_lagrange, f(omega^i)) == MSM(g, coeffs of f) == [f(tau)]G."""
    tau = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA % pyref.R
    params = z.kzg.ParamsKZG.setup(k, tau, backend=be)
    n = 1 << k
    g0 = orc.g1_affine_to_ints(params.g_host[:2])
    assert g0[0] == (1, 2) and g0[1] == pyref.g1_mul(pyref.G1_GEN, tau)
    evals = pc.rand_fr(orc, pyref, n, seed)
    coeffs = z.domain.EvaluationDomain(3, k, backend=be).lagrange_to_coeff(evals)
    c1, c2 = params.commit_lagrange(evals), params.commit(coeffs)
    assert (c1 == c2).all() and not (c1 == 0).all()
    f_tau = pyref.poly_eval(orc.fr_to_ints(coeffs), tau)
    want = orc.g1_to_affine(orc.g1_mul(orc.g1_generator(), orc.fr_from_ints([f_tau])[0]))[0]
    assert (c1[:8] == want).all()
    params.release()


def test_emulated_kzg_setup_consistency(emu, orc, pyref):
    _kzg_consistency(emu, orc, pyref, 4, 7)


@pytest.mark.gpu
@pytest.mark.parametrize("log_n", [0, 1, 6, 9])
def test_gpu_g1_ntt(gpu, orc, pyref, log_n):
    _check_vs_oracle(gpu, orc, pyref, log_n, seed=log_n)


@pytest.mark.gpu
def test_gpu_kzg_setup_consistency(gpu, orc, pyref):
    _kzg_consistency(gpu, orc, pyref, 12, 8)
