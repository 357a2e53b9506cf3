#!/usr/bin/env python3
"""bench.py — proofs/hour of the sgx_dcap_verifier k=19 prover on MI355X (BASELINE.json configs[1]) and, as extras, the
BN254 MSM Mscalar/s at 2^24 (configs[2]) and batched NTT (configs[3]).

Default (`--mode prove`): a "step" is one batch of `--inflight` (4) REAL proofs — `zk_plonk_create_proof` (the library's C++ create_proof, csrc/prover.hip;
`--prover python`: its twin zk-dcap-verifier_amd/plonk/prover.py, same bytes), one context + HIP stream + host thread per proof in flight; the K timed steps are
a continuous pipeline (every thread proves K proofs back to back, no join between steps; barrier + synchronize on both sides of the K steps) — over a satisfiable synthetic circuit with the census of the sgx circuit at
k = 19 (tools/sgx_shaped_circuit.py: 25 advice, 18 fixed, 11 lookups, 16 equality columns, 24 gates, degree 5 => 71 commitments,
64 iNTT(2^19), 64 NTT(2^21) + 1 iNTT(2^21), evaluate_h over 2^21 rows, 175 evaluations, SHPLONK).  The witness columns are
resident in HBM when the timed region starts (witness synthesis is host work the north star leaves in Rust); everything from the
advice commitments to the last SHPLONK commitment — transcript hashing on the host included — is inside it.  Outside the timed
region one proof per context is handed to the pure-Python verifier together with the CPU baseline (the reference's own
acceptance check, sgx_dcap_verifier.rs:826-844); a proof that does not verify fails the run.
`value` is quoted on the HBM-resident witness (bench contract).  The same K steps are then repeated with the witness starting in
HOST memory and crossing PCIe inside every proof (zk_dev_upload_batch, the proofs in flight taking turns on the link) — `extra.host_witness` (page-locked
staging memory / pageable arrays) — and the
boundary as INTEGRATION.md 2 first wires it (one blocking host-buffer call per best_multiexp / best_fft / evaluate_h, a13-a16 on the
CPU) is replayed as `extra.thin_shim`.  Extras also RESULT-check the BASELINE microbench sizes: MSM 2^20 / 2^24 closed form,
batched NTT 2^22 x 25 round trip + 64 outputs against the direct sum.

`--mode opmix`: the earlier hot-path op-mix (MSM / NTT / quotient call list of create_proof over synthetic columns, half
uniform, half witness-like sparse; no grand products / lookup permutation / evaluations / SHPLONK) — kept for continuity with
profiles/r01 run1-run27.

N > 1: one process per GPU, every rank proves its own stream of proofs (weak scaling, no data-path collective) — that is `value`.
Reported under "extra": ONE proof spread over the ranks (`sharded_proof`: SRS tables sharded by index range, 128-byte partial
commitments all-gathered; the quotient sharded by extended-domain coset, numerators all-gathered between HBM buffers over RCCL;
its bytes must equal the single-GPU proof) and the sharded MSM alone (`msm_sharded_2^{21,24}`).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def rand_fr(n, seed):
    """n field elements uniform in [0, r) by rejection (BASELINE.md §3 / SURVEY 8d cfg 3) as (n, 4) uint64 raw limbs, read as Montgomery forms."""
    from zk_dcap_verifier_amd.fields import rand_fr_array
    return rand_fr_array(np.random.default_rng(seed), n)


def witness_like(n, seed):
    """90 % zeros, 8 % bytes, 2 % uniform (SURVEY.md 8d cfg 2), canonical small values in Montgomery
    form would need a field multiplication; for the bucket distribution it is enough that the
    *canonical* value is small, so build canonical values and convert with the known R on the GPU side
    is unnecessary: zeros stay zeros; for the rest we place small canonical values * R mod r."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    vals = np.zeros((n, 4), dtype=np.uint64)
    small = np.nonzero((u >= 0.9) & (u < 0.98))[0]
    Rm = (1 << 256) % R_MOD
    table = np.array([[(b * Rm % R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for b in range(256)], dtype=np.uint64)
    vals[small] = table[rng.integers(0, 256, size=small.size)]
    big = np.nonzero(u >= 0.98)[0]
    vals[big] = rand_fr(big.size, seed + 1)
    return vals


def sgx_shaped_program(z, k, ek, A, F, L, n_perm, d):
    """Synthetic proving-key program with the census of the sgx circuit:
    degree-3 gates q*(a + b*c - d) on the Fp-chip columns, 4-5 expression lookups compressed with theta."""
    ev = z.evaluation
    one = np.array([(((1 << 256) % R_MOD) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    g = ev.Graph()
    g.add_constant(one)
    r = [g.add_rotation(x) for x in (0, 1, 2, 3)]
    gates = []
    for i in range(min(A, 24)):
        a, b, c, dd = (ev.vs(ev.ADVICE, i, r[j]) for j in range(4))
        t = g.add_calculation(ev.MUL, b, c)
        t = g.add_calculation(ev.ADD, a, t)
        t = g.add_calculation(ev.SUB, t, dd)
        gates.append(g.add_calculation(ev.MUL, ev.vs(ev.FIXED, i % F, r[0]), t))
    g.add_calculation(ev.HORNER, ev.vs(ev.PREVIOUS), gates, ev.vs(ev.Y))
    lookups = []
    for n in range(L):
        lg = ev.Graph()
        r0 = lg.add_rotation(0)
        m = 4 if n % 2 == 0 else 5
        sel = ev.vs(ev.FIXED, n % F, r0)
        ins = [lg.add_calculation(ev.MUL, sel, ev.vs(ev.ADVICE, (n + j) % A, r0)) for j in range(m)]
        ci = lg.add_calculation(ev.HORNER, ins[0], ins[1:], ev.vs(ev.THETA))
        ct = lg.add_calculation(ev.HORNER, ev.vs(ev.FIXED, (n + 1) % F, r0), [ev.vs(ev.FIXED, (n + 1 + j) % F, r0) for j in range(1, m)], ev.vs(ev.THETA))
        a1 = lg.add_calculation(ev.ADD, ci, ev.vs(ev.BETA))
        b1 = lg.add_calculation(ev.ADD, ct, ev.vs(ev.GAMMA))
        lg.add_calculation(ev.MUL, a1, b1)
        lookups.append(lg)
    return ev.Program(k=k, extended_k=ek, n_fixed=F, n_advice=A, n_instance=0, n_challenges=0, blinding_factors=5, cs_degree=d,
                      perm_columns=[(0, i % A) for i in range(n_perm)], custom_gates=g, lookups=lookups)


class ProofWorkload:
    def __init__(self, z, be, k, A, F, L, n_perm, d):
        self.z, self.be, self.k, self.A, self.F, self.L, self.d = z, be, k, A, F, L, d
        self.n = 1 << k
        ek = k
        while (1 << ek) < self.n * (d - 1):
            ek += 1
        self.ek, self.en = ek, 1 << ek
        chunk = d - 2
        self.P = (n_perm + chunk - 1) // chunk
        self.n_perm = n_perm
        n = self.n
        # SRS-shaped base tables: g, g_lagrange = n points with random discrete logs (fixed-base mul on GPU)
        self.tables = []
        for seed in (20241008, 20241009):
            ks = be.to_device(rand_fr(n, seed))
            pts = be.alloc(n * 64)
            be.g1_fixed_base_mul(ks, n, pts)
            self.tables.append(be.bases_register((pts, n)))
            ks.free()
            pts.free()
        self.g, self.g_lagrange = self.tables
        # resident columns (Lagrange form): half uniform, half witness-like sparse
        ncol = A + 3 * L + self.P
        self.cols = [be.to_device(rand_fr(n, 100 + i) if i % 2 == 0 else witness_like(n, 100 + i)) for i in range(ncol)]
        self.work = [be.alloc(n * 32) for _ in range(ncol)]            # per-proof copies that the iNTT overwrites
        self.hpoly = be.to_device(rand_fr(n, 7))
        # extended cosets: advice + perm z + 3 per lookup are produced per proof; fixed / sigma / l* are pk data
        self.ext_dyn = [be.alloc(self.en * 32) for _ in range(ncol)]
        # proving-key cosets: distinct HBM buffers (no aliasing, so cache reuse is not flattered),
        # filled on the GPU from one uploaded random column scaled by distinct constants
        seed_col = be.to_device(rand_fr(self.en, 300))
        mults = rand_fr(F + n_perm + 3, 301)

        def derived(i):
            b = be.alloc(self.en * 32)
            be.fr_scale_dev(seed_col, mults[i], b, self.en)
            return b
        self.ext_fixed = [derived(i) for i in range(F)]
        self.ext_sigma = [derived(F + i) for i in range(n_perm)]
        self.ext_l = [derived(F + n_perm + i) for i in range(3)]
        seed_col.free()
        self.h_ext = be.alloc(self.en * 32)
        self.prog = sgx_shaped_program(z, k, ek, A, F, L, n_perm, d)
        self.evaluator = z.evaluation.Evaluator(self.prog, backend=be)
        self.scal = rand_fr(4, 9)
        self.n_msm = A + 3 * L + self.P + 1 + (d - 1) + 2
        self.n_intt = ncol
        self.n_ext = ncol

    def step(self):
        be, n, k, ek = self.be, self.n, self.k, self.ek
        A, L, P = self.A, self.L, self.P
        lib, ctx = be.lib, be.ctx
        # commitments, batched per proof phase exactly where create_proof's transcript allows it
        # (all columns of a phase are committed before the next challenge is squeezed):
        nA, nP, nL = self.A, self.P, self.L
        adv, zs, lk = self.cols[:nA], self.cols[nA:nA + nP], self.cols[nA + nP:]
        be.msm_batch(self.g_lagrange, adv, n)                           # phase 2: advice columns
        be.msm_batch(self.g_lagrange, lk[nL:3 * nL], n)                 # phase 3: permuted input / table of every lookup
        be.msm_batch(self.g_lagrange, zs + lk[:nL], n)                  # phase 4: permutation and lookup grand products
        be.msm(self.g, self.hpoly, n)                                   # phase 5: vanishing argument's random poly
        # phase 6: lagrange -> coeff -> extended coset
        for c, w in zip(self.cols, self.work):
            be.fr_scale_dev(c, self.scal[0], w, n)                      # per-proof copy (blinding changes every proof)
        be.lagrange_to_coeff_batch_dev(self.work, k)                    # all committed columns of the proof
        be.coeff_to_extended_batch_dev(self.work, self.ext_dyn, k, ek)
        adv = self.ext_dyn[:A]
        zs = self.ext_dyn[A:A + P]
        lk = self.ext_dyn[A + P:]
        self.evaluator.evaluate_h(fixed=self.ext_fixed, advice=adv, instance=[], l0=self.ext_l[0], l_last=self.ext_l[1], l_active_row=self.ext_l[2],
                                  perm_cosets=self.ext_sigma, perm_products=zs, lookup_product=lk[0:L], lookup_input=lk[L:2 * L],
                                  lookup_table=lk[2 * L:3 * L], challenges=[], beta=self.scal[0], gamma=self.scal[1], theta=self.scal[2],
                                  y=self.scal[3], out=self.h_ext)
        # phase 7: h = numerator / (X^n - 1), back to coefficients, commit the d-1 pieces
        be.divide_by_vanishing_poly_dev(self.h_ext, k, ek)
        be.extended_to_coeff_dev(self.h_ext, k, ek)
        be.msm_batch(self.g, [self.h_ext.ptr + i * n * 32 for i in range(self.d - 1)], n)
        # phase 9: SHPLONK h(X) and linearisation commitments
        be.msm(self.g, self.work[0], n)
        be.msm(self.g, self.work[1], n)



TAU = 0x1C59A59B6CFF4308740943526ADE1D8C09F71B337A67269CC89586BCDD6DFCBA   # SRS trapdoor of the synthetic setup
PROVER = "native"      # --prover: "native" = zk_plonk_create_proof (C++ over the C ABI), "python" = its twin plonk.create_proof


class ProverWorkload:
    """keygen once, then one create_proof per step on a copy of the resident witness."""

    def __init__(self, z, be, k, circuit, srs=None, pk=None):
        self.z, self.be, self.k, self.n = z, be, k, 1 << k
        cs, fixed, asm, advice = circuit
        # one SRS and one proving key for the process: the first context runs ParamsKZG::setup (fixed-base powers + EC-NTT) and keygen; the others share its
        # expanded tables, its columns and its compiled programs in HBM (zk_bases_share, zk_quotient_program_share)
        self.params = z.kzg.ParamsKZG.setup(k, TAU, backend=be) if srs is None else z.kzg.ParamsKZG.shared_with(srs, be)
        self.pk = z.plonk.keygen(self.params, cs, fixed, asm) if pk is None else z.plonk.ProvingKey.shared_with(pk, be)
        self.native = z.plonk.NativeProver(self.params, self.pk) if PROVER == "native" else None
        self.master = [be.to_device(a) for a in advice]
        self.work = [be.alloc(self.n * 32) for _ in advice]
        self.advice_host = advice                      # pageable host arrays (what a Rust Vec<Fr> is)
        self.pinned = None                             # page-locked copies (zk_host_alloc), made on first use
        self.seed = 0
        self.proof, self.info = None, None
        dom = self.pk.domain
        self.ek, self.en, self.d = dom.extended_k, dom.extended_n, cs.degree()
        L, A = len(cs.lookups), cs.num_advice_columns
        chunk = cs.permutation_chunk_len()
        self.P = (len(cs.permutation_columns) + chunk - 1) // chunk
        self.A, self.F, self.L, self.n_perm = A, cs.num_fixed_columns, L, len(cs.permutation_columns)
        self.n_msm = A + 3 * L + self.P + 1 + (self.d - 1) + 2
        self.n_intt = A + 3 * L + self.P
        self.n_ext = self.n_intt

    upload_turn = threading.Lock()

    def step(self, timings=None, witness="resident", capture=None):
        """witness = "resident": the advice columns are in HBM when the step starts (a device-to-device copy, create_proof works in place);
        "pinned" / "pageable": they start in host memory and cross PCIe inside the step (page-locked staging memory / ordinary memory)."""
        from zk_dcap_verifier_amd.transcript import Blake2bWrite
        if witness == "resident":
            for w, m in zip(self.work, self.master):
                w.copy_from(m)
            adv = self.work
        else:
            if witness == "pinned" and self.pinned is None:
                self.pinned = [self.be.host_alloc((self.n, 4)) for _ in self.advice_host]
                for p_, a_ in zip(self.pinned, self.advice_host):
                    p_[:] = a_
            # the PCIe hop of the witness, inside the step: one zk_dev_upload_batch of the proof's columns.  The proofs in flight take turns on the link
            # (one upload at a time), so in steady state one proof's upload runs under the kernels of the others instead of all uploads colliding.
            with ProverWorkload.upload_turn:
                self.be.upload_columns(self.work, self.pinned if witness == "pinned" else self.advice_host, self.n * 32)
            adv = self.work
        self.seed += 1
        if self.native is not None and capture is None:
            # the native per-proof path (zk_plonk_create_proof, csrc/prover.hip): the C++ twin of plonk.create_proof — same bytes, no interpreter in the loop
            self.proof = self.native.create_proof(adv, [], np.random.default_rng(self.seed))
            if timings is not None:
                timings.update(self.native.phase_ms)           # zk_plonk_last_phase_ms: the driver's own wall clock per phase
            if self.info is None:
                n_commit = self.A + 3 * self.L + self.P + 1 + (self.d - 1) + 2
                self.info = {"commitments": n_commit, "evals": len(self.proof) // 32 - n_commit}
            return
        tr = Blake2bWrite()
        self.info = self.z.plonk.create_proof(self.params, self.pk, adv, [], np.random.default_rng(self.seed), tr, timings=timings, capture=capture)
        self.proof = tr.finalize()


class ThinShimReplay:
    """INTEGRATION.md §2 taken literally: halo2's create_proof stays on the CPU and ONLY best_multiexp / best_fft / evaluate_h are redirected, one
    blocking host-buffer call each (zk_msm, zk_ntt, zk_evaluate_h): no phase batching, every column crosses PCIe per call, and a13-a16 (lookup
    permutation, grand products, evaluations, SHPLONK combinations) remain CPU loops.  This replays the call list of ONE real proof (columns
    captured from plonk.create_proof) through those entry points and sums the time spent inside them; the CPU loops are timed by the CPU leg on the
    port (oracle/) and added there.  It is the measurement of the boundary as a maintainer would first wire it, next to `value` (the phase-batched,
    HBM-resident prover)."""

    def __init__(self, z, be, wl):
        self.z, self.be, self.wl = z, be, wl
        cap = {}
        wl.step(capture=cap)
        self.cap = cap
        n, k = wl.n, wl.k
        pk = wl.pk
        dl = lambda d: d.download((n, 4))
        dom = z.domain.EvaluationDomain(wl.d, k, backend=be)
        bf = pk.vk.cs.blinding_factors()
        one = fr_mont_limbs(1)
        l0 = np.zeros((n, 4), dtype=np.uint64); l0[0] = one
        ll = np.zeros((n, 4), dtype=np.uint64); ll[n - bf - 1] = one
        la = np.zeros((n, 4), dtype=np.uint64); la[: n - bf - 1] = one
        self.ev = z.evaluation.Evaluator(pk.program, backend=be)
        self.ev.load_pk([dl(d) for d in pk.fixed_polys], [dl(d) for d in pk.sigma_polys], dom.lagrange_to_coeff(l0), dom.lagrange_to_coeff(ll),
                        dom.lagrange_to_coeff(la))
        self.sigma_values = [dl(d) for d in pk.sigma_values]
        self.fixed_values = [dl(d) for d in pk.fixed_values]
        self.w_inv = fr_mont_limbs(pow(pow(7, (R_MOD - 1) >> k, R_MOD), -1, R_MOD))
        self.scratch = np.empty((n, 4), dtype=np.uint64)

    def step(self):
        be, wl, cap, n, k = self.be, self.wl, self.cap, self.wl.n, self.wl.k
        gl, g = wl.params.g_lagrange.handle, wl.params.g.handle
        t = {"zk_msm": 0.0, "zk_ntt": 0.0, "zk_evaluate_h": 0.0}
        cnt = {"zk_msm": 0, "zk_ntt": 0, "zk_evaluate_h": 0}

        def msm(h, col):
            t0 = time.perf_counter(); be.msm(h, col, n); t["zk_msm"] += time.perf_counter() - t0; cnt["zk_msm"] += 1
        perm = [c for pr in cap["permuted"] for c in pr]
        for col in cap["advice"] + perm + cap["perm_products"] + cap["lookup_products"]:
            msm(gl, col)
        msm(g, cap["random_poly"])
        polys = []
        for col in cap["advice"] + cap["perm_products"] + cap["lookup_products"] + perm:       # lagrange_to_coeff = best_fft(omega_inv) + CPU scaling
            a = col.copy()
            t0 = time.perf_counter(); be.ntt(a, k, self.w_inv); t["zk_ntt"] += time.perf_counter() - t0; cnt["zk_ntt"] += 1
            polys.append(a)
        nA, nZ, L = len(cap["advice"]), len(cap["perm_products"]), len(cap["lookup_products"])
        fm = fr_mont_limbs
        t0 = time.perf_counter()
        h = self.ev.evaluate_h_polys(advice=polys[:nA], instance=[], perm_products=polys[nA:nA + nZ], lookup_product=polys[nA + nZ:nA + nZ + L],
                                     lookup_input=polys[nA + nZ + L:][0::2], lookup_table=polys[nA + nZ + L:][1::2], challenges=[],
                                     beta=fm(cap["beta"]), gamma=fm(cap["gamma"]), theta=fm(cap["theta"]), y=fm(cap["y"]), finish=True)
        t["zk_evaluate_h"] += time.perf_counter() - t0; cnt["zk_evaluate_h"] += 1
        for i in range(wl.d - 1):
            msm(g, h[i * n:(i + 1) * n])
        msm(g, polys[0]); msm(g, polys[1])                               # SHPLONK's two commitments
        return t, cnt

    def release(self):
        self.ev.release()


class PhaseBatchedReplay:
    """The integration level BETWEEN the thin shim and `value`: shim/halo2_proofs_mi355x/src/prover_phases.patch — an otherwise unchanged plonk/prover.rs whose
    commitment loops, transforms and argument provers each became ONE batch of device calls per phase, while halo2's Polynomial values stay host Vec<Fr>s.  Every
    phase therefore uploads the witness-dependent columns it reads and downloads the columns it produces (pageable host memory, as a Vec is); only the proving key
    stays in HBM; the transcript runs on the host.  Measured by running the Python twin — the phase list of that patch, call for call — with the two hooks that move
    those bytes for real (plonk/prover.py `phase_io`); the proof bytes are unchanged and compared.  `cache` = the variant with a pointer-keyed device cache in the
    shim: a column that is still in HBM from the phase that uploaded or produced it is not uploaded again (downloads stay: halo2 owns host copies)."""

    def __init__(self, z, be, wl, cache):
        self.z, self.be, self.wl, self.cache = z, be, wl, cache
        self.host, self.bytes_up, self.bytes_down, self.ms, self.skipped_s = {}, 0, 0, {}, 0.0

    def _key(self, col):
        from zk_dcap_verifier_amd._lib import _dptr
        return _dptr(col)

    def reads(self, phase, cols):
        import ctypes as C
        be, nb = self.be, self.wl.n * 32
        t0 = time.perf_counter()
        for col in cols:
            kptr = self._key(col)
            if self.cache and kptr in self.host:
                continue                                               # still resident: the shim's cache hands back the device copy
            a = self.host.get(kptr)
            if a is None:
                # the advice columns at their first re-read: halo2 holds them on the host WITH their blinding rows (this replay lets the library write those on the
                # device, so the host copy is fetched here) — bookkeeping of the replay, outside its clock and its byte counts
                t_skip = time.perf_counter()
                a = np.empty((self.wl.n, 4), dtype=np.uint64)
                be._ck(be.lib.zk_dev_download(be.ctx, a.ctypes.data_as(C.c_void_p), C.c_void_p(kptr), C.c_size_t(nb)))
                self.host[kptr] = a
                self.skipped_s += time.perf_counter() - t_skip
                t0 += time.perf_counter() - t_skip
                if self.cache:
                    continue
            be._ck(be.lib.zk_dev_upload(be.ctx, C.c_void_p(kptr), a.ctypes.data_as(C.c_void_p), C.c_size_t(nb)))
            self.bytes_up += nb
        self.ms[phase + ":upload"] = self.ms.get(phase + ":upload", 0.0) + (time.perf_counter() - t0) * 1e3

    def writes(self, phase, cols):
        import ctypes as C
        be, nb = self.be, self.wl.n * 32
        t0 = time.perf_counter()
        for col in cols:
            kptr = self._key(col)
            a = self.host.get(kptr)
            if a is None:
                a = self.host[kptr] = np.empty((self.wl.n, 4), dtype=np.uint64)
            be._ck(be.lib.zk_dev_download(be.ctx, a.ctypes.data_as(C.c_void_p), C.c_void_p(kptr), C.c_size_t(nb)))
            self.bytes_down += nb
        self.ms[phase + ":download"] = self.ms.get(phase + ":download", 0.0) + (time.perf_counter() - t0) * 1e3

    def step(self):
        from zk_dcap_verifier_amd.transcript import Blake2bWrite
        wl = self.wl
        self.host, self.bytes_up, self.bytes_down, self.ms, self.skipped_s = {}, 0, 0, {}, 0.0
        for w, m in zip(wl.work, wl.master):
            w.copy_from(m)
        wl.be.sync()
        tr = Blake2bWrite()
        t0 = time.perf_counter()
        # phase 2 of a host-resident prover: the advice columns cross PCIe inside commit_lagrange_batch (pageable memory)
        wl.be.upload_columns(wl.work, wl.advice_host, wl.n * 32)
        self.bytes_up += len(wl.work) * wl.n * 32

        timings = {}
        self.z.plonk.create_proof(wl.params, wl.pk, wl.work, [], np.random.default_rng(wl.seed), tr, timings=timings, phase_io=self)
        ms = (time.perf_counter() - t0 - self.skipped_s) * 1e3
        return ms, tr.finalize(), dict(self.ms), timings, self.bytes_up, self.bytes_down


def cpu_a13_a16(orc, shim, threads):
    """The CPU side of the thin shim: the loops of create_proof that the three redirected functions do not cover, on the
    port (oracle/bn254_oracle.c), over the captured columns of a real proof.  Independent items (the lookups, the queries) run on a thread pool as
    halo2 spreads them over rayon workers; the port's loops themselves are single-threaded C."""
    from concurrent.futures import ThreadPoolExecutor
    cap, wl = shim.cap, shim.wl
    cs = wl.pk.vk.cs
    k, n, bf = wl.k, wl.n, cs.blinding_factors()
    fm = lambda x: orc.fr_from_ints([x])[0]
    beta, gamma = fm(cap["beta"]), fm(cap["gamma"])
    blind1 = rand_fr(bf + 1, 5)
    out = {}
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        t = time.time()
        list(pool.map(lambda c: orc.lookup_permute(c[0], c[1], k, bf, blind1, blind1), cap["compressed"]))
        out["permute_expression_pair_ms"] = (time.time() - t) * 1e3
        t = time.time()
        list(pool.map(lambda q: orc.lookup_product(q[0][0], q[0][1], q[1][0], q[1][1], k, beta, gamma, blind1[:bf]), zip(cap["compressed"], cap["permuted"])))
        out["lookup_commit_product_ms"] = (time.time() - t) * 1e3
        t = time.time()
        cols = {0: cap["advice"], 1: shim.fixed_values}
        vals = [cols[ty][i] for ty, i in cs.permutation_columns]
        chunk = cs.permutation_chunk_len()
        DELTA = pow(7, 1 << 28, R_MOD)
        # halo2 walks the sets one after the other (each starts at the previous set's last value) and spreads every set's row loops over rayon; the
        # port's row loop is serial, so the sets run side by side here instead (the chaining is one scalar multiplication per row: timing-neutral)
        list(pool.map(lambda s0: orc.permutation_product(vals[s0:s0 + chunk], shim.sigma_values[s0:s0 + chunk], k, beta, gamma,
                                                         fm(pow(DELTA, s0, R_MOD)), fm(1), blind1[:bf]), range(0, len(vals), chunk)))
        out["permutation_commit_ms"] = (time.time() - t) * 1e3
        t = time.time()
        pt = fm(cap["y"])
        n_eval = wl.info["evals"] + 1
        list(pool.map(lambda i: orc.eval_polynomial(cap["advice"][i % len(cap["advice"])], pt), range(n_eval)))
        out["eval_polynomial_ms"] = (time.time() - t) * 1e3
        # SHPLONK: every committed polynomial enters two linear combinations (its rotation set's Q_i and L(X)): one scale + one add over n
        # coefficients each, then one kate_division per opening point of each set and one for L(X)
        t = time.time()
        n_polys = wl.n_msm - 2 + wl.F + wl.n_perm
        a0 = cap["advice"][0]

        def muladd(_):
            orc.fr_add(orc.fr_mul(a0, a0), a0)
        list(pool.map(muladd, range(2 * n_polys)))
        list(pool.map(lambda _: orc.kate_division(a0, pt), range(8)))
        out["shplonk_combinations_ms"] = (time.time() - t) * 1e3
    out = {k_: round(v, 1) for k_, v in out.items()}
    out["total_ms"] = round(sum(out.values()), 1)
    out["threads"] = threads
    out["what"] = ("C port of the CPU loops a thin shim leaves in halo2 (permute_expression_pair, lookup / permutation grand products, eval_polynomial per query, "
                   "SHPLONK combinations + kate_division) on the captured columns of one k = 19 proof; theta-compression of the lookup expressions not counted")
    return out


def cpu_baseline(cfg, threads, program_for=None, verify=None, ntt_checks=(), shim=None):
    """The CPU leg: (1) the oracle ("port": C restatement of halo2's CPU algorithms, oracle/bn254_oracle.c) timed on a bounded
    sample of the proof's MSM / NTT / evaluate_h calls and extrapolated to one proof; (2) with `verify` = (vk, tau, instances,
    proof): the reference's acceptance check — verify_proof (pure-Python, oracle/verifier.py) on a proof the GPU just produced.
    Checker code, never the thing shipped or measured."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    k, ek = cfg["k"], cfg["ek"]
    n = 1 << k
    sc = rand_fr(n, 1)
    bases = orc.gen_bases_arith(5, 3, n, threads=threads)

    def median3(f):                                                   # one call of a primitive is noisy on a shared 256-thread host: the median of three
        ts = []
        for _ in range(3):
            t = time.time(); f(); ts.append(time.time() - t)
        return sorted(ts)[1]
    t_msm = median3(lambda: orc.best_multiexp(sc, bases, threads=threads))
    w = orc.Domain(cfg["d"], k)
    t_intt = median3(lambda: w.lagrange_to_coeff(sc, threads=threads))
    t_ext = median3(lambda: w.coeff_to_extended(sc, threads=threads))
    # quotient on 2^17 rows of the same program shape, scaled by rows
    import zk_dcap_verifier_amd as z
    ks = min(k, 17 - (ek - k))
    prog = program_for(ks, ks + (ek - k)) if program_for else sgx_shaped_program(z, ks, ks + (ek - k), cfg["A"], cfg["F"], cfg["L"], cfg["n_perm"], cfg["d"])
    size = 1 << (ks + ek - k)
    col = rand_fr(size, 3)
    P = (cfg["n_perm"] + cfg["d"] - 3) // (cfg["d"] - 2)
    blob = prog.to_blob()
    t_q = median3(lambda: orc.evaluate_h(blob, [col] * cfg["F"], [col] * cfg["A"], [], col, col, col, [col] * cfg["n_perm"], [col] * P, [col] * cfg["L"], [col] * cfg["L"],
                                         [col] * cfg["L"], [], col[0], col[1], col[2], col[3], size, threads=threads)) * ((1 << ek) / size)
    total = cfg["n_msm"] * t_msm + cfg["n_intt"] * t_intt + (cfg["n_ext"] + 1) * t_ext + t_q
    out = {"value": round(3600.0 / total, 3), "unit": "proofs/hour", "cores": threads, "kind": "port",
           "sample": f"median of 3: best_multiexp(2^{k}) {t_msm:.2f}s x{cfg['n_msm']}, lagrange_to_coeff {t_intt:.3f}s x{cfg['n_intt']}, "
                     f"coeff_to_extended(2^{ek}) {t_ext:.3f}s x{cfg['n_ext'] + 1}, evaluate_h on 2^{ks + ek - k} rows scaled to 2^{ek} = {t_q:.2f}s "
                     "(grand products, lookup permutation, evaluations and SHPLONK of the CPU prover are NOT counted: the baseline is optimistic); "
                     "C restatement of halo2 CPU algorithms (pthreads), not the Rust binary"}
    if shim is not None:
        try:
            out["a13_a16_cpu_port"] = cpu_a13_a16(orc, shim, threads)
            # with the CPU loops of a13-a16 timed, the baseline proof is complete (round 1 left them out): fold them into `value`
            total += out["a13_a16_cpu_port"]["total_ms"] * 1e-3
            out["value"] = round(3600.0 / total, 3)
            out["sample"] = out["sample"].replace("(grand products, lookup permutation, evaluations and SHPLONK of the CPU prover are NOT counted: the baseline is optimistic)",
                                                  f"+ lookup permutation, grand products, evaluations and SHPLONK combinations on the port {out['a13_a16_cpu_port']['total_ms'] / 1e3:.2f}s")
        except Exception as e:
            out["a13_a16_cpu_port_error"] = repr(e)
    if ntt_checks:
        # SURVEY 8d cfg 4: outputs of the batched GPU NTT against the direct sum out[j] = sum_i a_i omega^(i j) (Horner at omega^j, oracle C)
        t = time.time()
        bad = 0
        for col, j, w, got in ntt_checks:
            want = orc.eval_polynomial(col, orc.fr_from_ints([pow(w, j, R_MOD)])[0])
            bad += 0 if (np.asarray(got) == want).all() else 1
        out["ntt_direct_sum_check"] = {"outputs": len(ntt_checks), "ok": bad == 0, "seconds": round(time.time() - t, 2)}
        if bad:
            raise RuntimeError(f"bench: {bad} of {len(ntt_checks)} batched-NTT outputs differ from the direct sum")
    if verify is not None:
        import verifier
        vk, tau, instances, proofs = verify
        t = time.time()
        oks = [bool(verifier.verify_proof(vk, tau, instances, pr)) for pr in proofs]
        out["verify_proof"] = {"accepted": all(oks), "proofs": len(oks), "seconds": round(time.time() - t, 2),
                               "what": "pure-Python plonk::verify_proof + VerifierSHPLONK on the GPU prover's proof bytes (oracle/verifier.py)"}
        if not all(oks):
            raise RuntimeError("bench: a proof produced by the GPU prover was REJECTED by verify_proof")
    return out


def dot_mod_r(a, b):
    """sum_i a_i * b_i mod r of two (n, 4) uint64 limb arrays (raw 256-bit values), exact and independent of the GPU library: 16-bit limbs
    as float64, 2^16 rows per BLAS product (every partial sum is an integer < 2^32 * 2^16 = 2^48 < 2^53, so the float64 sums are exact in
    any order), accumulated in uint64 and recombined with Python integers.  16 M terms take about a second."""
    a = np.ascontiguousarray(a, dtype="<u8").reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype="<u8").reshape(-1, 4)
    assert a.shape == b.shape
    blk = 1 << 16
    av, bv = a.view("<u2"), b.view("<u2")
    A, B = np.empty((blk, 16)), np.empty((blk, 16))
    acc, tot, pending = np.zeros((16, 16), dtype=np.uint64), 0, 0

    def flush():
        nonlocal tot, pending
        for i in range(16):
            for j in range(16):
                tot += int(acc[i, j]) << (16 * (i + j))
        acc[:] = 0
        pending = 0
    for s0 in range(0, a.shape[0], blk):
        m = min(blk, a.shape[0] - s0)
        np.copyto(A[:m], av[s0:s0 + m], casting="unsafe")
        np.copyto(B[:m], bv[s0:s0 + m], casting="unsafe")
        acc += (A[:m].T @ B[:m]).astype(np.uint64)
        pending += 1
        if pending == 4096:                                          # 4096 * 2^48 = 2^60: flush before uint64 could overflow
            flush()
    flush()
    return tot % R_MOD


def _ints(a):
    """(n, 4) uint64 limbs -> list of Python ints (the raw 256-bit values)."""
    b = np.ascontiguousarray(a, dtype="<u8").tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def msm_microbench(be, log_n, seed, reps=3, verify=False):
    """SURVEY 8d cfg 3: bases P_i = [k_i]G built on the GPU, uniform scalars.  verify: the closed form
    MSM(s, P) = [sum_i s_i k_i mod r] G, checked against the (independent) fixed-base path of the library."""
    n = 1 << log_n
    kh = rand_fr(n, seed)
    ks = be.to_device(kh)
    pts = be.alloc(n * 64)
    be.g1_fixed_base_mul(ks, n, pts)
    t = time.time(); h = be.bases_register((pts, n)); t_reg = time.time() - t
    pts.free()
    sc = rand_fr(n, seed + 1)
    ks.upload(sc)
    res = be.msm(h, ks, n)
    t = time.time()
    for _ in range(reps):
        be.msm(h, ks, n)
    dt = (time.time() - t) / reps
    out = {"log_n": log_n, "ms": round(dt * 1e3, 3), "Mscalar_per_s": round(n / dt / 1e6, 2), "hbm_frac_algorithmic": round(96 * n / dt / 1e9 / HBM_PEAK_GBS, 5),
           "table_expand_s": round(t_reg, 3)}
    # one more call with the library's HIP-event timers on: where the time goes (the window width is the library's choice for the table size: 16 bits below 2^22 points, 20 from there)
    be.timing(True)
    be.msm(h, ks, n)
    ph = {lab: be.timing_get(lab)[0] for lab in ("msm_sort", "msm_accumulate", "msm_reduce")}
    pairs = be.stat_get("msm_pairs")
    be.timing(False)
    out["phase_ms"] = {lab: round(v, 3) for lab, v in ph.items() if v is not None}
    if pairs:
        out["windows_per_scalar"] = round(pairs / n, 2)
    if verify:
        # the limb arrays are Montgomery forms: value = limbs * R^-1; sum_i s_i k_i = (sum_i S_i K_i) * R^-2, and the
        # fixed-base kernel takes a Montgomery scalar, so feed it (sum S_i K_i) * R^-1
        rinv = pow(1 << 256, -1, R_MOD)
        tot = dot_mod_r(kh, sc) * rinv % R_MOD
        one = be.to_device(np.array([[(tot >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64))
        o = be.alloc(64)
        be.g1_fixed_base_mul(one, 1, o)
        want = o.download((8,))
        out["closed_form_check"] = bool((res[:8] == want).all())
        one.free()
        o.free()
    be.bases_release(h)
    ks.free()
    return out


def fr_mont_limbs(x):
    return np.array([(((x % R_MOD) << 256) % R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def ntt_microbench(be, log_n, cols, seed, reps=3):
    """SURVEY 8d cfg 4: `cols` distinct columns of 2^log_n uniform field elements through ONE zk_ntt_batch_dev call per transform (the batched
    entry point create_proof uses), forward with omega then inverse.  Checked here: the round trip of two whole columns.  Returned for the CPU
    leg (the only place the oracle may be used): 64 (column, j, value) outputs to compare with the direct sum sum_i a_i omega^(i j)."""
    n = 1 << log_n
    w = pow(7, (R_MOD - 1) >> log_n, R_MOD)
    wm, wim, ninv = fr_mont_limbs(w), fr_mont_limbs(pow(w, -1, R_MOD)), fr_mont_limbs(pow(n, -1, R_MOD))
    keep = {0: rand_fr(n, seed), cols - 1: rand_fr(n, seed + cols - 1)}          # host copies of the checked columns
    dev = [be.to_device(keep[c] if c in keep else rand_fr(n, seed + c)) for c in range(cols)]
    be.ntt_batch_dev(dev, log_n, wm)
    rng = np.random.default_rng(seed)
    checks = []
    for t in range(64):
        c = (0, cols - 1)[t % 2]
        jj = int(rng.integers(0, n)) if t >= 4 else (0, n - 1, 1, n // 2)[t]
        checks.append((keep[c], jj, w, dev[c].download((1, 4), offset=jj * 32)[0]))
    be.ntt_batch_dev(dev, log_n, wim)
    ok = True
    for c in keep:
        be.fr_scale_dev(dev[c], ninv, dev[c], n)
        ok = ok and bool((dev[c].download((n, 4)) == keep[c]).all())
        dev[c].upload(keep[c])
    be.sync()
    t0 = time.time()
    for _ in range(reps):
        be.ntt_batch_dev(dev, log_n, wm)
    be.sync()
    dt = (time.time() - t0) / reps
    for d in dev:
        d.free()
    be.trim_pool()
    return {"columns": cols, "entry_point": "zk_ntt_batch_dev", "ms_per_batch": round(dt * 1e3, 3), "ms_per_column": round(dt / cols * 1e3, 3),
            "GB_per_s_algorithmic": round(64 * n * cols / dt / 1e9, 1), "hbm_frac": round(64 * n * cols / dt / 1e9 / HBM_PEAK_GBS, 4),
            "roundtrip_check": ok}, checks


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=19)
    ap.add_argument("--advice", type=int, default=25)
    ap.add_argument("--fixed", type=int, default=18)
    ap.add_argument("--lookups", type=int, default=11)
    ap.add_argument("--perm-columns", type=int, default=16)
    ap.add_argument("--degree", type=int, default=5)
    ap.add_argument("--mode", choices=("prove", "opmix"), default="prove", help="prove: real create_proof over the sgx-shaped circuit (default); opmix: the hot-path call list over synthetic columns")
    ap.add_argument("--census", choices=("chip_estimate", "reference_exact"), default="chip_estimate",
                    help="which synthetic circuit `value` is measured on (tools/sgx_shaped_circuit.py); the other one is measured as extra.census_* unless --no-extras")
    ap.add_argument("--prover", choices=("native", "python"), default="native",
                    help="which create_proof the timed steps run: the library's C++ one (zk_plonk_create_proof) or its Python twin (plonk.create_proof); same bytes")
    ap.add_argument("--no-jit", action="store_true", help="the headline key's quotient on the micro-op interpreter instead of kernels generated for its program (tune quot_jit)")
    ap.add_argument("--inflight", type=int, default=4, help="proofs processed concurrently per step on one GPU (one context + HIP stream each)")
    ap.add_argument("--no-extras", action="store_true", help="skip the MSM 2^24 / NTT 2^22 microbenchmarks and the CPU baseline")
    args = ap.parse_args(argv)

    global PROVER
    PROVER = args.prover
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("ZK_BENCH_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        # RCCL over xGMI on a multi-GPU node; gloo only for the CPU plumbing test / a single-GPU dry run of the N > 1 path
        backend = os.environ.get("ZK_BENCH_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    else:
        backend = "none"
    tdev = "cuda" if backend == "nccl" else "cpu"
    if os.environ.get("ZK_SWITCH_INTERVAL"):                 # experiment knob: CPython's GIL hand-over interval for the per-proof host threads
        sys.setswitchinterval(float(os.environ["ZK_SWITCH_INTERVAL"]))
    import zk_dcap_verifier_amd as z
    lib_path = os.environ.get("ZK_LIB") or None             # A/B: another build of the product library on the same box (same ABI)
    be = z.Backend(local, lib_path)            # raises if the HIP library / GPU is missing: no CPU path
    assert "gfx950" in be.version() or os.environ.get("ZK_BENCH_PLUMBING_TEST") == "1"

    # One step = one batch of `inflight` proofs, each on its own context/stream and host thread, so the
    # latency-bound phases of one proof (bucket reduction, scans) overlap the throughput-bound phases of another.
    import threading
    inflight = max(1, args.inflight)
    bes = [be] + [z.Backend(local, lib_path) for _ in range(inflight - 1)]
    bench_tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("ZK_TUNE", "").split(",") if kv}   # experiment knob: zk_tune_set on every context
    for b in bes:
        if bench_tune:
            b.tune(**bench_tune)
    if args.mode == "prove":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import sgx_shaped_circuit as sgx
        circuit = sgx.build(z, be, args.k, census=args.census)  # one satisfying witness, shared by the contexts
        wls = []
        # the headline key's quotient runs on kernels GENERATED for its program (csrc/quotient_jit.hip: hiprtc at key build, tune quot_jit; the contexts that borrow the key
        # borrow the kernels).  Only this key: the other keys a default run builds (second census, configs[0], sharded extras) keep the interpreter — a key build costs
        # ~13 s of compilation (level 1: the degree parts) on a box that has not seen the program before (comgr's cache makes it 0.1 s afterwards).  --no-jit / ZK_TUNE=quot_jit=0: interpreter everywhere.
        want_jit = not args.no_jit and bench_tune.get("quot_jit", 1) != 0 and not os.environ.get("ZK_BENCH_PLUMBING_TEST")
        if want_jit:
            be.tune(quot_jit=1)
        for b in bes:
            wls.append(ProverWorkload(z, b, args.k, circuit, srs=wls[0].params if wls else None, pk=wls[0].pk if wls else None))
            if b is be:
                jit_compile_s = be.stat_get("quot_jit_compile_s")
                be.tune(quot_jit=bench_tune.get("quot_jit", 0) if "quot_jit" in bench_tune and not want_jit else 0)
        quotient_executor = {"generated_kernels": bool(want_jit and jit_compile_s > 0), "hiprtc_compile_s": round(jit_compile_s, 2),
                             "what": "the headline key's quotient program as straight-line kernels generated for it at key build (csrc/quotient_jit.hip, tune quot_jit = 1); "
                                     "false = the micro-op interpreter (csrc/quotient.hip)"}
    else:
        circuit = None
        wls = [ProofWorkload(z, b, args.k, args.advice, args.fixed, args.lookups, args.perm_columns, args.degree) for b in bes]
    wl = wls[0]

    def step_all(steps=1, **kw):
        """`steps` steps = steps x inflight proofs: every host thread proves `steps` proofs back to back on its own context, with no join between steps —
        a continuous pipeline, so the proofs drift out of phase and the GPU never waits for the slowest proof of a batch"""
        if inflight == 1:
            for _ in range(steps):
                wl.step(**kw)
            return
        errors = []

        def loop(w):
            try:
                for _ in range(steps):
                    w.step(**kw)
            except BaseException as e:
                errors.append(e)
        ths = [threading.Thread(target=loop, args=(w,)) for w in wls]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errors:
            raise errors[0]

    def barrier():
        for b in bes:
            b.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()

    step_all(args.warmup)
    for b in bes:
        b.timing(True)                         # HIP events on the library's own streams, inside the timed region
    barrier()
    t0 = time.time()
    step_all(args.steps)
    barrier()
    dt = time.time() - t0
    def tsum(label):
        ms = n = 0
        for b in bes:
            m, k_ = b.timing_get(label)
            ms += m or 0.0
            n += k_
        return ms, n
    acc_ms, acc_n = tsum("msm_accumulate")
    sort_ms, _ = tsum("msm_sort")
    red_ms, _ = tsum("msm_reduce")
    q_ms, q_n = tsum("quotient")
    ntt_s_ms, ntt_s_n = tsum("ntt_strided_pass")
    ntt_f_ms, ntt_f_n = tsum("ntt_final_pass")
    be_stats = {"msm_columns": sum(b.stat_get("msm_columns") for b in bes), "msm_pairs": sum(b.stat_get("msm_pairs") for b in bes),
                "ntt_points": sum(b.stat_get("ntt_points") for b in bes), "ntt_pass_points": sum(b.stat_get("ntt_pass_points") for b in bes),
                "quotient_alg_bytes": sum(b.stat_get("quotient_alg_bytes") for b in bes)}
    for b in bes:
        b.timing(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    proofs_total = args.steps * inflight
    proofs_per_hour = world * 3600.0 * proofs_total / dt

    extra = {"ops_per_proof": {"msm": wl.n_msm, "intt_2^k": wl.n_intt, "ntt_2^ek": wl.n_ext + 1, "quotient_rows": wl.en},
             # NOT kernel time: the span between a HIP event pair on one stream while the other proofs' streams share the GPU (a kernel waits for CUs other streams hold);
             # kernel time per proof = extra.single_proof.kernel_ms (a proof alone on the GPU) and profiles/*_kernel_trace_per_proof_inflight1.txt
             "event_span_ms_per_proof_pipelined": {"msm_sort": round(sort_ms / proofs_total, 3), "msm_accumulate": round(acc_ms / proofs_total, 3),
                                                   "msm_reduce": round(red_ms / proofs_total, 3), "quotient": round(q_ms / proofs_total, 3) if q_ms else None} if inflight > 1 else None,
             "proofs_in_flight": inflight, "ms_per_proof": round(dt / proofs_total * 1e3, 3)}
    if args.mode == "prove":
        extra["quotient_executor"] = quotient_executor
    # roofline of the dominant kernel (msm_accumulate).  Algorithmic bytes = 96 B per (scalar, base) pair
    # (SURVEY 8d); one launch covers a whole batch, so bytes/launch = 96 * n * columns-per-launch.
    msm_columns = be_stats["msm_columns"]
    msm_pairs = be_stats["msm_pairs"]
    acc_s = max(acc_ms * 1e-3, 1e-9)                              # (the CPU plumbing test has no event timing)
    achieved = 96.0 * wl.n * msm_columns / acc_s / 1e9
    # HBM bytes per launch from the PMC counters: they cannot be collected inside this run (separate rocprofv3 --pmc passes, MI355X_MICROARCH.md), so the
    # figure is the one of the last profiling session (tools/collect_profiles.sh + tools/summarize_profiles.py), cited with its source
    # Two readings of the same counters are given, because the guide's x2 on FETCH_SIZE is calibrated for wide coalesced streaming reads only and this kernel's reads
    # are 64-byte gathers (one table point per pair): `raw` = (FETCH_SIZE + WRITE_SIZE) KB x 1024 as counted, `doubled_fetch` = (2 FETCH_SIZE + WRITE_SIZE) x 1024; the
    # kernel's own design traffic (`expected_gather_bytes`: 64 B per pair + its 4-byte reference) lies next to `raw`, which is the reading that applies to gathers
    # (tools/microbench `gather64` under --pmc FETCH_SIZE calibrates it: DESIGN.md 5).
    traffic, traffic_source, tjd = None, None, {}
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        try:
            tjd = json.load(open(tj))
            traffic_source = tjd.get("source", "").split(" (")[0]
            kern = tjd.get("kernels", {}).get("msm_accumulate_kernel")
            if kern:
                traffic = {"raw": round(kern["fetch_bytes_per_launch"] + kern["write_bytes_per_launch"]),
                           "doubled_fetch": round(2 * kern["fetch_bytes_per_launch"] + kern["write_bytes_per_launch"])}
            elif tjd.get("msm_accumulate_bytes_per_launch"):           # (a traffic.json of rounds 1-3: only the doubled reading was kept)
                traffic = {"raw": None, "doubled_fetch": tjd.get("msm_accumulate_bytes_per_launch")}
        except Exception:
            traffic = None
    # The bucket step in the kernel that ran: on carry-free 29-bit limbs (the only form since round 4: 1467 v_mad_u64_u32, no addc, ~650 other VALU instructions per
    # mixed addition — the kernel's own ISA census) or round 2's 32-bit redundant form (1160 mad + addc pairs + ~700 others).
    limb29 = True
    # register-loop rate of the same step (tools/microbench, 4 blocks per CU: profiles/r03/run93_microbench_mad_issue_rate.txt) ...
    XYZZ_MADD_PEAK = 17.70e9 if limb29 else 13.84e9
    # ... and an integer bound that owes nothing to this repo's loops: instruction ISSUE rates measured on the box — v_mad_u64_u32 with 12 independent accumulators
    # per lane 55.2 lane-ops per clock per CU (the 39.6 of rounds 1-2 came from a latency-limited four-chain loop), simple VALU 114.6 — times the step's instruction census, at the
    # 2.4 GHz nominal clock.  Under these kernels the chip runs at 1.8-2.0 GHz (roofline.int_alu.pmc.eff_clock_ghz), which is most of the distance to this bound.
    MAD_RATE, SIMPLE_RATE, N_CU, CLK = 55.2, 114.6, 256, 2.4e9
    step_cycles = (1467 / MAD_RATE + 650 / SIMPLE_RATE) if limb29 else (1160 / MAD_RATE + (1160 + 700) / SIMPLE_RATE)
    ISSUE_BOUND = N_CU * CLK / step_cycles
    valu = {}
    try:
        valu = json.load(open(tj)).get("valu", {}) if os.path.exists(tj) else {}
    except Exception:
        valu = {}
    acc_valu = valu.get("msm_accumulate_kernel", {})
    roofline = {"kernel": "msm_accumulate_kernel", "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                "avg_launch_ms": round(acc_ms / max(acc_n, 1), 4), "launches": acc_n,
                "algorithmic_bytes_per_launch": round(96.0 * wl.n * msm_columns / max(acc_n, 1)),
                "note": "the kernel is integer-ALU bound (v_mad_u64_u32), not HBM bound - DESIGN.md 3.2; int_alu gives the fraction of the "
                        "measured XYZZ mixed-add peak; with several proofs in flight a launch's event time includes cycles shared with other streams' kernels, so both fractions are lower bounds",
                "int_alu": {"achieved_Gmadd_per_s": round(msm_pairs / acc_s / 1e9, 3), "peak_Gmadd_per_s": XYZZ_MADD_PEAK / 1e9,
                            "frac": round(msm_pairs / acc_s / XYZZ_MADD_PEAK, 4),
                            "issue_bound_Gmadd_per_s": round(ISSUE_BOUND / 1e9, 3), "issue_bound_frac": round(msm_pairs / acc_s / ISSUE_BOUND, 4),
                            "issue_bound_is": ("256 CUs x 2.4 GHz / (1467 v_mad_u64_u32 / 55.2 + 650 other VALU / 114.6 lane-ops per clk per CU): the 29-bit-limb step's instruction census at the issue rates tools/microbench measured" if limb29 else
                                               "256 CUs x 2.4 GHz / (1160 v_mad_u64_u32 / 55.2 + 1860 v_addc + other VALU / 114.6): the 32-bit step's instruction census at the issue rates tools/microbench measured"),
                            "step": "xyzz29_madd_fast (9 x 29-bit limbs)" if limb29 else "xyzz_madd_lazy (8 x 32-bit limbs)",
                            # counters of the last profiling session (separate rocprofv3 --pmc passes, one proof alone): VALU busy share, per-wave issue / stall split,
                            # the chip's effective clock under this kernel and the instruction-mix issue model (tools/summarize_profiles.py valu_section)
                            "pmc": {k_: acc_valu.get(k_) for k_ in ("valu_busy", "active_valu_per_wave_cycle", "wait_inst_per_wave_cycle", "wait_any_per_wave_cycle", "eff_clock_ghz", "int64_share", "issue_model")} if acc_valu else None,
                            "pmc_source": json.load(open(tj)).get("valu_source") if acc_valu else None},
                # the other throughput-bound kernels, same counters (profiles/traffic.json): NTT passes and the quotient interpreter
                "other_kernels_pmc": {k_: {c_: valu[k_].get(c_) for c_ in ("ms_per_launch", "valu_busy", "active_valu_per_wave_cycle", "wait_inst_per_wave_cycle", "eff_clock_ghz", "int64_share", "issue_model")}
                                      for k_ in ("ntt_strided_pass_kernel", "ntt_strided_pass29_kernel", "ntt_final_pass_kernel", "quotient_kernel", "zkq_generated") if k_ in valu} or None}
    if isinstance(traffic, dict):
        alg = roofline["algorithmic_bytes_per_launch"]
        pairs_per_launch = msm_pairs / max(acc_n, 1)
        traffic["expected_gather_bytes"] = round(pairs_per_launch * 68)    # one 64-byte table point + its 4-byte sorted reference per point addition: the kernel's design traffic
        traffic["ratio_to_algorithmic"] = {"raw": round(traffic["raw"] / alg, 2) if traffic.get("raw") and alg else None,
                                           "doubled_fetch": round(traffic["doubled_fetch"] / alg, 2) if traffic.get("doubled_fetch") and alg else None}
        traffic["applies"] = "raw (64-byte gathers; the x2 of MI355X_MICROARCH.md is for wide coalesced streaming reads)"
    # Every throughput-bound kernel class with its own line: algorithmic bytes of SURVEY 8d over the kernel's HIP-event time INSIDE the timed
    # region (event pairs on the library's streams; with several proofs in flight a pair also spans cycles given to other streams' kernels, so these are lower bounds —
    # `alone` repeats them with one proof in flight when the extras run).
    def kline(name, alg_bytes, ms, launches, **more):
        ach = alg_bytes / max(ms * 1e-3, 1e-9) / 1e9 if ms else None
        return dict({"kernel": name, "bound": "hbm", "algorithmic_bytes": round(alg_bytes), "ms": round(ms, 3) if ms else None, "launches": launches,
                     "achieved": round(ach, 2) if ach else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5) if ach else None}, **more)
    ntt_ms = (ntt_s_ms or 0.0) + (ntt_f_ms or 0.0)
    passes = be_stats["ntt_pass_points"] / be_stats["ntt_points"] if be_stats["ntt_points"] else None
    roofline["kernels"] = [
        kline("msm_accumulate_kernel", 96.0 * wl.n * msm_columns, acc_ms, acc_n, per_unit="96 B per (scalar, base) pair", int_alu_frac=roofline["int_alu"]["issue_bound_frac"]),
        kline("ntt_strided_pass29_kernel + ntt_final_pass_kernel", 64.0 * be_stats["ntt_points"], ntt_ms, ntt_s_n + ntt_f_n, per_unit="64 B per point per transform",
              passes=round(passes, 2) if passes else None, hbm_bytes_moved_estimate=round(64.0 * be_stats["ntt_pass_points"]),
              strided_ms=round(ntt_s_ms or 0.0, 3), final_ms=round(ntt_f_ms or 0.0, 3)),
        kline("quotient (kernels generated for the key's program)" if args.mode == "prove" and quotient_executor["generated_kernels"] else "quotient_kernel",
              be_stats["quotient_alg_bytes"], q_ms or 0.0, q_n, per_unit="(columns + 1) x 32 B per evaluated row")]

    if args.mode == "prove":
        # The same K steps once more with the witness starting in HOST memory (what the Rust boundary hands over: create_proof receives host-owned
        # circuits, sgx_dcap_verifier.rs:814-822): 25 advice columns = n * 32 B each cross PCIe inside every proof, from page-locked staging
        # memory (zk_host_alloc) and, with extras, from ordinary pageable arrays.  `value` stays the HBM-resident figure (bench contract); these
        # are the PCIe-inclusive rates.
        host_rates = {}
        for kind in (("pinned",) if args.no_extras else ("pinned", "pageable")):
            step_all(1, witness=kind)                                # untimed: page-locks / first touch
            barrier()
            t1 = time.time()
            step_all(args.steps, witness=kind)
            barrier()
            dh = time.time() - t1
            if dist is not None:
                th_ = torch.tensor([dh], dtype=torch.float64, device=tdev)
                dist.all_reduce(th_, op=dist.ReduceOp.MAX)
                dh = float(th_.item())
            host_rates[kind] = {"proofs_per_hour": round(world * 3600.0 * proofs_total / dh, 2), "ms_per_proof": round(dh / proofs_total * 1e3, 3)}
        extra["host_witness"] = dict(host_rates, bytes_per_proof=wl.A * wl.n * 32,
                                     what="same steps with the advice columns starting in host memory: every proof uploads its columns first (one zk_dev_upload_batch, the proofs in "
                                          "flight take turns on the link), then create_proof runs on the device copies; the upload of one proof overlaps the kernels of the others")
    if args.mode == "prove" and not args.no_extras and world == 1:
        # The OTHER census of the synthetic circuit, same K steps, same contexts: "reference_exact" builds the base64 part exactly as
        # the reference configures and assigns it (15 advice columns that are zero outside 1696 rows, 7 lookups of 4/5 expressions on 65-/257-row
        # tables) next to the chip estimate; "chip_estimate" guesses every column.  The real circuit's cost lies wherever its un-vendored chips put it.
        other = "reference_exact" if args.census == "chip_estimate" else "chip_estimate"
        census2 = None
        try:
            circuit2 = sgx.build(z, be, args.k, census=other)
            wls2 = []
            for b in bes:
                wls2.append(ProverWorkload(z, b, args.k, circuit2, srs=wl.params, pk=wls2[0].pk if wls2 else None))

            def step2(steps):
                def loop(w_):
                    for _ in range(steps):
                        w_.step()
                ths_ = [threading.Thread(target=loop, args=(w_,)) for w_ in wls2]
                for t_ in ths_:
                    t_.start()
                for t_ in ths_:
                    t_.join()
            step2(1)
            barrier()
            t1 = time.time()
            step2(args.steps)
            barrier()
            d2 = time.time() - t1
            extra["census_" + other] = {"proofs_per_hour": round(3600.0 * proofs_total / d2, 2), "ms_per_proof": round(d2 / proofs_total * 1e3, 3),
                                         "commitments": wls2[0].info["commitments"], "proof_bytes": len(wls2[0].proof),
                                         "shape": f"A={wls2[0].A} F={wls2[0].F} L={wls2[0].L} perm_columns={wls2[0].n_perm} degree={wls2[0].d}"}
            census2 = (wls2[0].pk.vk, [w_.proof for w_ in wls2])       # verified by the CPU leg
            for w_ in reversed(wls2):                                  # (borrowers of the shared key first, its owner last)
                for d_ in w_.master + w_.work:
                    d_.free()
                w_.pk.release()
        except Exception as e:
            extra["census_" + other + "_error"] = repr(e)
    if args.mode == "prove" and not args.no_extras and world == 1:
        # BASELINE configs[0]: the shape of the reference's stack-B circuit (crates/p256-ecdsa, k = 18, snark-verifier's Poseidon transcript: base.rs:200-212) — the case the
        # reference itself runs on the CPU.  Degree 4: three h pieces, so the key holds three cosets of its columns and h(X) comes from zk_cosets_to_pieces_dev.
        try:
            import p256_shaped_circuit as p256
            kb = 18 if args.k >= 18 else max(args.k - 1, 5)           # (the plumbing test runs the whole file at a tiny k)
            cs_b, fixed_b, asm_b, advice_b, inst_b = p256.build(kb)
            params_b = z.kzg.ParamsKZG.setup(kb, TAU, backend=be)
            pk_b = z.plonk.keygen(params_b, cs_b, fixed_b, asm_b)
            prover_b = z.plonk.NativeProver(params_b, pk_b, transcript="poseidon")
            dev_b = [be.to_device(a_) for a_ in advice_b]
            work_b = [be.alloc(a_.nbytes) for a_ in advice_b]
            times_b, proof_b = [], None
            for r_ in range(6):
                for w_, m_ in zip(work_b, dev_b):
                    w_.copy_from(m_)
                be.sync()
                t1 = time.time()
                proof_b = prover_b.create_proof(work_b, inst_b, np.random.default_rng(r_))
                times_b.append(time.time() - t1)
            best = sorted(times_b[1:])[len(times_b[1:]) // 2]
            extra["cfg1_p256_k18"] = {"ms_per_proof_alone": round(best * 1e3, 2), "proofs_per_hour_one_at_a_time": round(3600.0 / best, 1), "proof_bytes": len(proof_b),
                                      "pieces_from_cosets": bool(pk_b.pieces_from_cosets), "transcript": "poseidon", "shape": f"k={kb}, 3 advice, 1 lookup, 15 instance values, degree 4 (3 h pieces)",
                                      "what": "BASELINE configs[0] (the reference's own CPU-runnable case) through zk_plonk_create_proof; one proof at a time — at this size a proof is latency, not throughput"}
            cfg1 = (pk_b.vk, inst_b, proof_b)
            for d_ in dev_b + work_b:
                d_.free()
            pk_b.release()
            params_b.release()
        except Exception as e:
            extra["cfg1_p256_k18_error"] = repr(e)
    if args.mode == "prove" and not args.no_extras:
        # latency of ONE proof with the GPU to itself (the timed region above measures throughput with several in flight)
        lat = []
        for _ in range(2):
            tm = {}
            be.sync()
            t1 = time.time()
            wl.step(timings=tm)
            lat.append((time.time() - t1, tm))
        best = min(lat, key=lambda x: x[0])
        extra["single_proof"] = {"ms": round(best[0] * 1e3, 2), "phase_ms": {k_: round(v, 2) for k_, v in best[1].items()},
                                 "side_lane": "on (prover_side_lane = 1: a proof that is alone runs each phase's transforms on the context's helper context beside its commitments, DESIGN 3.7)"}
        if PROVER == "native" and inflight >= 2 and world == 1:
            # ... and what ONE and TWO proving threads give (the headline takes `inflight` of them): back-to-back proofs, same contexts
            def few(nth, reps):
                def loop_(w_):
                    for _ in range(reps):
                        w_.step()
                ths_ = [threading.Thread(target=loop_, args=(w_,)) for w_ in wls[:nth]]
                barrier()
                t_ = time.time()
                for t__ in ths_:
                    t__.start()
                for t__ in ths_:
                    t__.join()
                barrier()
                return nth * reps / (time.time() - t_) * 3600.0
            try:
                extra["proofs_per_hour_by_proving_threads"] = {"1": round(few(1, 6), 1), "2": round(few(2, 5), 1), str(inflight): round(proofs_per_hour, 1)}
            except Exception as e:
                extra["proofs_per_hour_by_proving_threads_error"] = repr(e)
        # the same proof once more with HIP-event timing on: alone on the GPU — and with the side lane OFF, so that no two kernels of the proof overlap — the event pairs
        # bracket only this proof's kernels, so these (unlike event_span_ms_per_proof_pipelined above) are kernel times
        be.tune(prover_side_lane=0)
        t1 = time.time()
        wl.step()
        extra["single_proof"]["ms_without_side_lane"] = round((time.time() - t1) * 1e3, 2)
        be.timing(True)
        wl.step()
        be.tune(prover_side_lane=bench_tune.get("prover_side_lane", 1))
        alone = {lab: be.timing_get(lab) for lab in ("msm_sort", "msm_accumulate", "msm_reduce", "quotient", "ntt_strided_pass", "ntt_final_pass")}
        pairs_alone = be.stat_get("msm_pairs")
        alone_stats = {lab: be.stat_get(lab) for lab in ("ntt_points", "ntt_pass_points", "quotient_alg_bytes")}
        be.timing(False)
        # ... and the NTT / quotient lines of roofline.kernels with the GPU to this one proof
        n_ms = (alone["ntt_strided_pass"][0] or 0.0) + (alone["ntt_final_pass"][0] or 0.0)
        for kl in roofline["kernels"]:
            if kl["kernel"].startswith("ntt") and n_ms:
                kl["alone"] = {"ms": round(n_ms, 3), "achieved": round(64.0 * alone_stats["ntt_points"] / (n_ms * 1e-3) / 1e9, 2),
                               "frac": round(64.0 * alone_stats["ntt_points"] / (n_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
            if kl["kernel"] == "quotient_kernel" and alone["quotient"][0]:
                kl["alone"] = {"ms": round(alone["quotient"][0], 3), "achieved": round(alone_stats["quotient_alg_bytes"] / (alone["quotient"][0] * 1e-3) / 1e9, 2),
                               "frac": round(alone_stats["quotient_alg_bytes"] / (alone["quotient"][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
        extra["single_proof"]["kernel_ms"] = {lab: round(v[0], 3) for lab, v in alone.items() if v[0] is not None}
        if alone["msm_accumulate"][0]:
            extra["single_proof"]["int_alu_frac_alone"] = round(pairs_alone / (alone["msm_accumulate"][0] * 1e-3) / XYZZ_MADD_PEAK, 4)
            # the same roofline with the kernel alone on the GPU: in the timed region three proofs share the chip, so an event pair around a launch also
            # spans cycles given to other streams' kernels (rocprofv3 sees the same stretched durations)
            a_ms, a_n = alone["msm_accumulate"]
            ach = 96.0 * wl.n * wl.n_msm / (a_ms * 1e-3) / 1e9
            roofline["alone"] = {"avg_launch_ms": round(a_ms / max(a_n, 1), 4), "launches": a_n, "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5),
                                 "what": "one proof with the GPU to itself (extra.single_proof): the kernel's own duration"}
            roofline["kernels"][0]["alone"] = {"ms": round(a_ms, 3), "achieved": round(ach, 2), "frac": round(ach / HBM_PEAK_GBS, 5)}
    cfg = {"k": args.k, "ek": wl.ek, "A": wl.A, "F": wl.F, "L": wl.L, "n_perm": wl.n_perm, "d": wl.d,
           "n_msm": wl.n_msm, "n_intt": wl.n_intt, "n_ext": wl.n_ext}
    cpu = None
    ntt_checks = []
    census2 = locals().get("census2")
    if rank == 0 and not args.no_extras and world == 1:
        try:
            extra["msm_2^20"] = msm_microbench(be, 20, 20241008, verify=True)
            extra["msm_2^21"] = msm_microbench(be, 21, 20241012, verify=True)      # the size of BASELINE configs[4]: the N = 1 point of extra.msm_sharded_2^21's curve
            extra["msm_2^24"] = msm_microbench(be, 24, 20241010, reps=2, verify=True)
        except Exception as e:
            extra["msm_microbench_error"] = str(e)
        try:
            if globals().get("PLUMBING_SKIP_NTT22"):
                raise RuntimeError("skipped (plumbing test)")
            extra["ntt_2^22_x25"], ntt_checks = ntt_microbench(be, 22, 25, 20241011)
        except Exception as e:
            extra["ntt_microbench_error"] = str(e)
        shim = None
        if args.mode == "prove":
            try:
                shim = ThinShimReplay(z, be, wl)
                shim.step()
                reps = [shim.step() for _ in range(2)]
                tt_ = min(reps, key=lambda r: sum(r[0].values()))
                extra["thin_shim"] = {"gpu_call_ms": {k_: round(v * 1e3, 2) for k_, v in tt_[0].items()}, "calls": tt_[1],
                                      "gpu_calls_total_ms": round(sum(tt_[0].values()) * 1e3, 2),
                                      "what": "INTEGRATION.md 2 as written: one blocking HOST-buffer call per best_multiexp / best_fft / evaluate_h of a real proof's "
                                              "call list (no phase batching, columns cross PCIe per call); a13-a16 stay CPU loops (timed by the CPU leg below)"}
            except Exception as e:
                extra["thin_shim_error"] = repr(e)
            try:
                rungs = {}
                for name, cache in (("host_columns", False), ("host_columns_with_device_cache", True)):
                    pb = PhaseBatchedReplay(z, be, wl, cache)
                    pb.step()
                    ms_, proof_, io_, ph_, up_, down_ = min((pb.step() for _ in range(2)), key=lambda r: r[0])
                    ref_tr = __import__("zk_dcap_verifier_amd.transcript", fromlist=["Blake2bWrite"]).Blake2bWrite()
                    for w_, m_ in zip(wl.work, wl.master):
                        w_.copy_from(m_)
                    z.plonk.create_proof(wl.params, wl.pk, wl.work, [], np.random.default_rng(wl.seed), ref_tr)
                    worst = max(io_, key=io_.get) if io_ else None
                    rungs[name] = {"ms_per_proof": round(ms_, 1), "proofs_per_hour": round(3600e3 / ms_, 1), "same_bytes_as_the_resident_prover": proof_ == ref_tr.finalize(),
                                   "pcie_GB_up": round(up_ / 1e9, 2), "pcie_GB_down": round(down_ / 1e9, 2), "pcie_bytes_up": up_, "pcie_bytes_down": down_, "transfer_ms": {k_: round(v, 1) for k_, v in sorted(io_.items())},
                                   "transfer_ms_total": round(sum(io_.values()), 1), "largest_transfer": worst, "phase_ms_incl_transfers": {k_: round(v, 1) for k_, v in ph_.items()}}
                extra["phase_batched_shim"] = dict(rungs, what="shim/halo2_proofs_mi355x/src/prover_phases.patch: every phase of create_proof is ONE batch of device calls, halo2's Polynomial "
                                                   "values stay host Vec<Fr>s (pageable): each phase uploads the witness-dependent columns it reads and downloads the ones it produces; the "
                                                   "proving key is resident, the transcript on the host, one proof at a time (a CPU prover's loop); measured on the Python twin with real "
                                                   "transfers (plonk/prover.py phase_io).  The rungs: extra.thin_shim (one blocking call per best_multiexp / best_fft / evaluate_h) < this < "
                                                   "`value` (zk_plonk_prove: columns never leave HBM)")
            except Exception as e:
                extra["phase_batched_shim_error"] = repr(e)
        # the CPU leg (the only part of this file that may touch oracle/): baseline timing, verify_proof on the GPU's proofs, and the
        # direct-sum check of the NTT outputs kept above
        if args.mode == "prove":
            cpu = cpu_baseline(cfg, os.cpu_count() or 1, program_for=lambda kk, ee: z.plonk.compile_program(wl.pk.vk.cs, kk, ee),
                               verify=(wl.pk.vk, TAU, [], [w_.proof for w_ in wls]), ntt_checks=ntt_checks, shim=shim)
            if census2 is not None:
                import importlib
                vmod = importlib.import_module("verifier")               # oracle/verifier.py (sys.path set by cpu_baseline): CPU leg, checker only
                ok2 = all(bool(vmod.verify_proof(census2[0], TAU, [], pr_)) for pr_ in census2[1][:1])
                extra["census_" + ("reference_exact" if args.census == "chip_estimate" else "chip_estimate")]["verify_proof_accepted"] = ok2
                if not ok2:
                    raise RuntimeError("bench: the second-census proof was REJECTED by verify_proof")
            cfg1 = locals().get("cfg1")
            if cfg1 is not None:
                import importlib
                vmod = importlib.import_module("verifier")
                pmod = importlib.import_module("poseidon_ref")           # oracle/: the second writing of snark-verifier's Poseidon transcript (checker only)
                ok1 = bool(vmod.verify_proof(cfg1[0], TAU, cfg1[1], cfg1[2], reader=pmod.Reader))
                extra["cfg1_p256_k18"]["verify_proof_accepted"] = ok1
                if not ok1:
                    raise RuntimeError("bench: the p256-shaped k = 18 proof was REJECTED by verify_proof")
            if shim is not None and "thin_shim" in extra and isinstance(cpu, dict) and "a13_a16_cpu_port" in cpu:
                cpu_ms = cpu["a13_a16_cpu_port"]["total_ms"]
                tot = extra["thin_shim"]["gpu_calls_total_ms"] + cpu_ms
                extra["thin_shim"].update(cpu_a13_a16_ms=cpu_ms, ms_per_proof=round(tot, 1), proofs_per_hour=round(3600e3 / tot, 1))
            if shim is not None:
                shim.release()
        else:
            try:
                cpu = cpu_baseline(cfg, os.cpu_count() or 1, ntt_checks=ntt_checks)
            except Exception as e:  # the baseline is a report, never a dependency of the measurement
                cpu = {"error": str(e)}
        if ntt_checks and isinstance(cpu, dict) and "ntt_direct_sum_check" in cpu:
            extra["ntt_2^22_x25"]["direct_sum_check_64_outputs"] = cpu["ntt_direct_sum_check"]["ok"]
    def multi_gpu_extras():
        if tdev == "cuda":
            torch.cuda.set_device(local)                       # the current device is per host thread
        if world > 1 and not args.no_extras:
            # MSM with the base table sharded over the ranks (SURVEY 8d cfg 5 / 8e): rank g holds bases and scalars
            # [g*N/G, (g+1)*N/G); the 128-byte XYZZ partials are all-gathered over RCCL/xGMI and summed on every rank.
            sizes = [int(x) for x in os.environ.get("ZK_BENCH_SHARDED_LOG_N", "21,24").split(",")]
            for logn in sizes:
                try:
                    n_loc = (1 << logn) // world
                    kh = rand_fr(n_loc, 31 + rank)
                    ks = be.to_device(kh)
                    pts = be.alloc(n_loc * 64)
                    be.g1_fixed_base_mul(ks, n_loc, pts)
                    h = be.bases_register((pts, n_loc))
                    pts.free()
                    sh = rand_fr(n_loc, 77 + rank)
                    ks.upload(sh)
                    gather = [torch.zeros(16, dtype=torch.int64, device=tdev) for _ in range(world)]

                    def sharded():
                        part = be.msm_partial(h, ks, n_loc)
                        mine = torch.from_numpy(part.view(np.int64).copy()).to(tdev)
                        dist.all_gather(gather, mine)
                        parts = torch.stack(gather).cpu().numpy().view(np.uint64)
                        return be.g1_sum_xyzz(parts)
                    res = sharded()
                    barrier()
                    t = time.time()
                    for _ in range(3):
                        sharded()
                    barrier()
                    ds = (time.time() - t) / 3
                    rec = {"ranks": world, "ms": round(ds * 1e3, 3), "Mscalar_per_s": round((1 << logn) / ds / 1e6, 2)}
                    if logn <= 21:   # closed form across ranks: sum over all shards of s_i k_i, checked via the fixed-base path
                        rinv = pow(1 << 256, -1, R_MOD)
                        loc = sum(a_ * b_ for a_, b_ in zip(_ints(kh), _ints(sh))) % R_MOD
                        lt = torch.tensor([(loc >> (62 * i)) & ((1 << 62) - 1) for i in range(5)], dtype=torch.int64, device=tdev)
                        allp = [torch.zeros(5, dtype=torch.int64, device=tdev) for _ in range(world)]
                        dist.all_gather(allp, lt)
                        tot = sum(sum(int(v) << (62 * i) for i, v in enumerate(p_.cpu().tolist())) for p_ in allp) % R_MOD * rinv % R_MOD
                        one = be.to_device(np.array([[(tot >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]], dtype=np.uint64))
                        o = be.alloc(64)
                        be.g1_fixed_base_mul(one, 1, o)
                        rec["closed_form_check"] = bool((res[:8] == o.download((8,))).all())
                        one.free()
                        o.free()
                    extra[f"msm_sharded_2^{logn}"] = rec
                    be.bases_release(h)
                    ks.free()
                except Exception as e:
                    extra[f"msm_sharded_2^{logn}_error"] = str(e)

        if world > 1 and not args.no_extras and args.mode == "prove":
            # ONE proof spread over the ranks (BASELINE configs[4] / SURVEY 8e): both SRS tables sharded by index range, every commitment of
            # create_proof = per-rank partial MSMs + one all_gather of 128-byte points; everything else is computed redundantly on every rank.
            try:
                def all_gather_points(part):                       # setup only (keygen's commitments of the sharded key): 128-byte points through torch
                    mine = torch.from_numpy(np.ascontiguousarray(part).view(np.int64).copy()).to(tdev)
                    out = [torch.zeros_like(mine) for _ in range(world)]
                    dist.all_gather(out, mine)
                    return torch.stack(out).cpu().numpy().view(np.uint64)
                sp = z.kzg.ParamsKZG.sharded(args.k, wl.params.g_host, wl.params.g_lagrange_host, rank, world, all_gather_points, backend=be)
                cs_, fixed_, asm_, _adv = circuit
                spk = z.plonk.keygen(sp, cs_, fixed_, asm_)
                # the per-proof path is the NATIVE prover in shard mode (zk_plonk_create_proof, csrc/prover.hip): its collective is the callback below — RCCL
                # all_gather_into_tensor between two HBM tensors the library uses as its exchange buffers (no host hop on the data path)
                n_cosets = 1 << (spk.domain.extended_k - args.k)
                units = sp.my_units(n_cosets)
                slots = -(-(n_cosets * sp.quotient_parts(n_cosets)) // world)
                cap = max(slots * (units[0][2] if units else wl.n) * 32, 1 << 16)
                # the exchange buffers must be memory the LIBRARY can address: HBM tensors on a GPU box (whatever the process group is), host tensors under the emulator
                lib_dev = "cuda" if (torch.cuda.is_available() and os.environ.get("ZK_BENCH_PLUMBING_TEST") != "1") else "cpu"
                xch = z.plonk.native.TorchExchange(world, cap, lib_dev, sync=be.sync, stage_through_host=(lib_dev == "cuda" and backend != "nccl"))
                snative = z.plonk.NativeProver(sp, spk, exchange=xch)
                phase_ms = {}

                def sharded_proof(seed):
                    for w_, m_ in zip(wl.work, wl.master):
                        w_.copy_from(m_)
                    pr_ = snative.create_proof(wl.work, [], np.random.default_rng(seed))
                    phase_ms.clear()
                    phase_ms.update(snative.phase_ms)
                    return pr_
                sharded_proof(1000)
                barrier()
                t = time.time()
                c0, b0 = xch.calls, xch.bytes
                for i_ in range(2):
                    pr_sharded = sharded_proof(1001 + i_)
                barrier()
                ds = (time.time() - t) / 2
                wl.seed = 1001                                     # the replica prover with the same RNG stream must emit the same bytes
                wl.step()
                tt = torch.tensor([ds, 1.0 if wl.proof == pr_sharded else 0.0], dtype=torch.float64, device=tdev)
                dist.all_reduce(tt, op=dist.ReduceOp.MIN)
                same = bool(tt[1].item() == 1.0)
                tt2 = torch.tensor([ds], dtype=torch.float64, device=tdev)
                dist.all_reduce(tt2, op=dist.ReduceOp.MAX)
                extra["sharded_proof"] = {"ranks": world, "prover": "native (zk_plonk_create_proof, shard mode)", "ms_per_proof": round(float(tt2.item()) * 1e3, 2),
                                          "identical_to_single_gpu_proof": same, "collective": f"{backend} all_gather_into_tensor between device tensors",
                                          "allgathers_per_proof": (xch.calls - c0) // 2, "allgather_bytes_per_rank_per_proof": (xch.bytes - b0) // 2,
                                          "phase_ms_rank0": {k_: round(v, 2) for k_, v in phase_ms.items()},
                                          "single_gpu_phase_ms": extra.get("single_proof", {}).get("phase_ms"),
                                          "what": "create_proof with both SRS tables sharded by index range (71 commitments = partial MSMs + all_gather of 128-byte XYZZ points, one per phase) "
                                                  "and the quotient sharded by extended-domain coset (size-n coset NTTs + evaluate_h per rank, one all_gather of the numerators)"}
                spk.release()
                sp.release()
            except Exception as e:
                extra["sharded_proof_error"] = str(e)


    # The N > 1 extras are the only collectives of a run that no earlier round could exercise on real RCCL: they run under a watchdog so that a rank that
    # raises (or a collective that never completes) costs the extras, not the bench line — `value` above is already measured.
    multi_hung = False
    if world > 1 and not args.no_extras:
        th_multi = threading.Thread(target=multi_gpu_extras, daemon=True)
        th_multi.start()
        th_multi.join(float(os.environ.get("ZK_BENCH_MULTI_TIMEOUT", "300")))
        if th_multi.is_alive():
            multi_hung = True
            extra["multi_gpu_extras_error"] = "timed out (a collective did not complete); the N > 1 extras were abandoned"

    if rank == 0:
        if args.mode == "prove":
            metric = "proofs/hour sgx_dcap_verifier k=19 (create_proof on the GPU: commitments, lookups, grand products, NTTs, evaluate_h, evaluations, SHPLONK, host transcript; witness synthesis excluded)"
            workload = (f"create_proof (halo2 mirror, zk-dcap-verifier_amd/plonk) of a satisfiable circuit with the sgx_dcap_verifier QE3-report census "
                        f"(tools/sgx_shaped_circuit.py): k={args.k}, extended_k={wl.ek}, A={wl.A} advice (14 full-width + 11 16-bit), F={wl.F} fixed, L={wl.L} lookups of 4-5 expressions, "
                        f"{wl.n_perm} equality columns (P={wl.P}), 24 gates, degree {wl.d}; per proof {wl.n_msm} MSM(2^{args.k}) + {wl.n_intt} iNTT + {wl.n_ext + 1} NTT(2^{wl.ek}) + evaluate_h "
                        f"+ {wl.info['evals'] + 1 if wl.info else '?'} evaluations + SHPLONK -> {len(wl.proof) if wl.proof else '?'}-byte proof; `value` = witness resident in HBM when a step starts (PCIe-inclusive rates: extra.host_witness; per-call host-buffer boundary: extra.thin_shim); "
                        f"one step = {inflight} proofs, {inflight} in flight on the GPU (one context + HIP stream + host thread each, proving back to back)")
        else:
            metric = "proofs/hour sgx_dcap_verifier k=19 (GPU hot path: MSM+NTT+quotient op-mix of create_proof; host witness/transcript excluded)"
            workload = (f"sgx_dcap_verifier QE3-report circuit shape, k={args.k}, extended_k={wl.ek}, A={args.advice} advice, F={args.fixed} fixed, "
                        f"L={args.lookups} lookups, {args.perm_columns} permutation columns (P={wl.P}), degree {args.degree}; "
                        f"{wl.n_msm} MSM(2^{args.k}) + {wl.n_intt} iNTT + {wl.n_ext + 1} NTT(2^{wl.ek}) + evaluate_h per proof; columns half uniform, half witness-like sparse; "
                        f"one step = a batch of {inflight} proofs in flight on the GPU (one HIP stream each)")
        line = {"metric": metric,
                "value": round(proofs_per_hour, 2), "unit": "proofs/hour", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x8 (256-bit Montgomery integers)",
                "data": "synthetic",
                "config": {"workload": workload, "mode": args.mode,
                           "prover": "zk_plonk_create_proof (C++, csrc/prover.hip)" if args.prover == "native" else "plonk.create_proof (Python twin)",
                           "parallelism": f"{world} x independent proofs (one process per GPU)",
                           "quotient": ("kernels generated for the key's program at key build (tune quot_jit = 1)" if args.mode == "prove" and quotient_executor["generated_kernels"] else "micro-op interpreter")},
                "roofline": roofline, "cpu_baseline": cpu, "extra": extra}
        print(json.dumps(line), flush=True)
    if multi_hung:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)                                            # a collective is stuck in a daemon thread: no orderly teardown is possible; the bench line is out (rank 0), the launcher sees the failure
    if dist is not None:
        dist.destroy_process_group()
    for b in bes:
        b.close()


if __name__ == "__main__":
    main()
