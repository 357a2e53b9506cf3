"""Parity cases shared by the emulator tests (CPU, tiny) and the GPU tests (through the product
C ABI).  Every case compares the backend with the oracle on the same seeded inputs, bit for bit."""
import random

import numpy as np

import zk_dcap_verifier_amd as z


def rand_fr(orc, pyref, n, seed):
    """n field elements uniform in [0, r) (rejection sampling), raw limbs read as Montgomery forms"""
    return np.ascontiguousarray(z.fields.rand_fr_array(np.random.default_rng(seed), n))


def check_vec_ops(be, orc, pyref, n, seed=11):
    a, b = rand_fr(orc, pyref, n, seed), rand_fr(orc, pyref, n, seed + 1)
    edge = orc.ints_to_limbs([0, 1, pyref.R - 1, pyref.R - 2, pyref.mont_r(pyref.R)])
    a[: len(edge)] = edge
    b[: len(edge)] = edge[::-1]
    da, db, do = be.to_device(a), be.to_device(b), be.alloc(a.nbytes)
    for name in ("mul", "add", "sub"):
        getattr(be, f"fr_{name}_dev")(da, db, do, n)
        assert (do.download(a.shape) == getattr(orc, f"fr_{name}")(a, b)).all(), name
    be.fr_scale_dev(da, b[7], do, n)
    assert (do.download(a.shape) == orc.fr_mul(a, np.repeat(b[7:8], n, axis=0))).all()
    aq = a.copy()
    aq[:2] = orc.ints_to_limbs([pyref.P - 1, pyref.R])   # valid Fq values that are not valid Fr values
    da.upload(aq)
    be.fq_mul_dev(da, db, do, n)
    assert (do.download(a.shape) == orc.fq_mul(aq, b)).all()
    for d in (da, db, do):
        d.free()


def msm_inputs(orc, pyref, n, seed, kind="uniform"):
    rnd = random.Random(seed)
    bases = orc.gen_bases_arith(rnd.randrange(1, pyref.R), rnd.randrange(1, pyref.R), n)
    if kind == "uniform":
        sc = rand_fr(orc, pyref, n, seed)
        if n > 8:
            sc[:4] = orc.fr_from_ints([0, 1, pyref.R - 1, 2])
    elif kind == "ones":
        sc = orc.fr_from_ints([1] * n)
    elif kind == "zeros":
        sc = orc.fr_from_ints([0] * n)
    elif kind == "witness":           # 90 % zeros, 8 % bytes, 2 % uniform (SURVEY 8d cfg 2)
        vals = []
        for _ in range(n):
            u = rnd.random()
            vals.append(0 if u < 0.9 else rnd.randrange(256) if u < 0.98 else rnd.randrange(pyref.R))
        sc = orc.fr_from_ints(vals)
    elif kind == "minus_one":
        sc = orc.fr_from_ints([pyref.R - 1] * n)
    else:
        raise ValueError(kind)
    return sc, bases


def check_msm(be, orc, pyref, n, seed=21, kind="uniform", repeat_bases=False, with_identity=False):
    sc, bases = msm_inputs(orc, pyref, n, seed, kind)
    if repeat_bases and n > 3:
        bases[1] = bases[0]
        bases[3] = bases[0]
        sc[1] = sc[0]                     # same scalar, same base -> the doubling branch of the mixed add
    if with_identity and n > 2:
        bases[2] = 0
    got = z.arithmetic.best_multiexp(sc, bases, backend=be)
    want = orc.g1_to_affine(orc.best_multiexp(sc, bases))[0]
    assert (got[:8] == want).all(), (n, kind)
    if (want == 0).all():
        assert (got[8:] == 0).all()
    else:
        assert orc.limbs_to_ints(got[8:])[0] == pyref.mont_r(pyref.P)      # z = mont(1)


def check_msm_prefix_and_handle(be, orc, pyref, n, seed=31):
    """commit() uses a prefix of the resident table; several MSMs reuse one registration."""
    sc, bases = msm_inputs(orc, pyref, n, seed)
    h = z.arithmetic.BasesHandle(be, bases)
    for m in (n, n - 1, max(1, n // 3), 1):
        got = z.arithmetic.best_multiexp(sc[:m], h)
        want = orc.g1_to_affine(orc.best_multiexp(sc[:m], bases[:m]))[0]
        assert (got[:8] == want).all(), m
    h.release()


def check_msm_batch(be, orc, pyref, n, count, seed=35, device=False):
    """zk_msm_batch: several columns (uniform, sparse, all ones, zeros ...) against one table."""
    kinds = ["uniform", "witness", "ones", "zeros", "minus_one"]
    cols, bases = [], None
    for i in range(count):
        sc, b = msm_inputs(orc, pyref, n, seed, kinds[i % len(kinds)] if i else "uniform")
        if i % len(kinds) == 0:
            sc = rand_fr(orc, pyref, n, seed + 13 * i)
        bases = b
        cols.append(sc)
    h = z.arithmetic.BasesHandle(be, bases)
    if device:
        dcols = [be.to_device(c) for c in cols]
        got = be.msm_batch(h.handle, dcols, n)
        for d in dcols:
            d.free()
    else:
        got = z.arithmetic.best_multiexp_batch(cols, h)
    for i, sc in enumerate(cols):
        want = orc.g1_to_affine(orc.best_multiexp(sc, bases))[0]
        assert (got[i, :8] == want).all(), i
        single = z.arithmetic.best_multiexp(sc, h)
        assert (single == got[i]).all(), i
    h.release()


def check_ntt(be, orc, pyref, log_n, seed=41):
    n = 1 << log_n
    a = rand_fr(orc, pyref, n, seed)
    w = orc.fr_from_ints([pyref.omega(log_n)])[0]
    b = a.copy()
    z.arithmetic.best_fft(b, w, log_n, backend=be)
    assert (b == orc.best_fft(a, w, log_n)).all(), log_n
    winv = orc.fr_from_ints([pow(pyref.omega(log_n), -1, pyref.R)])[0]
    z.arithmetic.best_fft(b, winv, log_n, backend=be)           # round trip: iNTT(NTT(a)) = n * a
    nm = orc.fr_from_ints([n % pyref.R])
    assert (b == orc.fr_mul(a, np.repeat(nm, n, axis=0))).all()


def check_domain_batch(be, orc, pyref, j, k, count, seed=55):
    """zk_lagrange_to_coeff_batch_dev / zk_coeff_to_extended_batch_dev / zk_ntt_batch_dev vs the oracle per column."""
    od = orc.Domain(j, k)
    n, en = 1 << k, 1 << od.extended_k
    cols = [rand_fr(orc, pyref, n, seed + i) for i in range(count)]
    dcols = [be.to_device(c) for c in cols]
    be.lagrange_to_coeff_batch_dev(dcols, k)
    coeffs = [od.lagrange_to_coeff(c) for c in cols]
    for d, want in zip(dcols, coeffs):
        assert (d.download((n, 4)) == want).all()
    outs = [be.alloc(en * 32) for _ in range(count)]
    be.coeff_to_extended_batch_dev(dcols, outs, k, od.extended_k)
    for o, c in zip(outs, coeffs):
        assert (o.download((en, 4)) == od.coeff_to_extended(c)).all()
    w = od.extended_omega
    be.ntt_batch_dev(outs, od.extended_k, w)
    for o, c in zip(outs, coeffs):
        assert (o.download((en, 4)) == orc.best_fft(od.coeff_to_extended(c), w, od.extended_k)).all()
    for d in dcols + outs:
        d.free()


def check_domain(be, orc, pyref, j, k, seed=51):
    d = z.domain.EvaluationDomain(j, k, backend=be)
    od = orc.Domain(j, k)
    assert d.extended_k == od.extended_k
    a = rand_fr(orc, pyref, 1 << k, seed)
    assert (d.lagrange_to_coeff(a) == od.lagrange_to_coeff(a)).all()
    ext = d.coeff_to_extended(a)
    assert (ext == od.coeff_to_extended(a)).all()
    assert (d.divide_by_vanishing_poly(ext) == od.divide_by_vanishing_poly(ext)).all()
    h = rand_fr(orc, pyref, 1 << d.extended_k, seed + 1)
    assert (d.extended_to_coeff(h) == od.extended_to_coeff(h)).all()
    back = d.extended_to_coeff(ext)                                # coeff -> extended -> coeff round trip
    assert (back[: 1 << k] == a).all() and (back[1 << k:] == 0).all()


def check_fixed_base(be, orc, pyref, n, seed=61):
    sc = rand_fr(orc, pyref, n, seed)
    sc[:3] = orc.fr_from_ints([0, 1, pyref.R - 1])
    ds, dout = be.to_device(sc), be.alloc(n * 64)
    be.g1_fixed_base_mul(ds, n, dout)
    got = dout.download((n, 8))
    g = orc.g1_generator()
    for i in list(range(min(n, 6))) + [n - 1]:
        want = orc.g1_to_affine(orc.g1_mul(g, sc[i]))[0]
        assert (got[i] == want).all(), i
    ds.free()
    dout.free()


def check_concurrent_callers(be, orc, pyref, n=200, threads=4):
    """The C ABI promises thread safety (rayon workers may call commit concurrently): hammer one context
    from several Python threads (ctypes releases the GIL) and check every result."""
    import threading
    sc, bases = msm_inputs(orc, pyref, n, 77)
    h = z.arithmetic.BasesHandle(be, bases)
    want_msm = orc.g1_to_affine(orc.best_multiexp(sc, bases))[0]
    log_n = 6
    a = rand_fr(orc, pyref, 1 << log_n, 78)
    w = orc.fr_from_ints([pyref.omega(log_n)])[0]
    want_ntt = orc.best_fft(a, w, log_n)
    errors = []

    def worker(i):
        try:
            for _ in range(3):
                if i % 2 == 0:
                    got = z.arithmetic.best_multiexp(sc, h)
                    assert (got[:8] == want_msm).all()
                else:
                    b = a.copy()
                    z.arithmetic.best_fft(b, w, log_n, backend=be)
                    assert (b == want_ntt).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    h.release()
    assert not errors, errors
