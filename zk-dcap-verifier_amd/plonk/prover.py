"""halo2_proofs::plonk::create_proof, MI355X edition — the caller of the whole hot path.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/plonk/prover.rs as the reference invokes it:
    create_proof::<KZGCommitmentScheme<Bn256>, ProverSHPLONK<Bn256>, Challenge255<G1Affine>, _, Blake2bWrite<..>, _>(
        &params, &pk, &[circuit], &[&[]], &mut rng, &mut transcript)          circuits/src/sgx_dcap_verifier.rs:814-822
Phase order, transcript traffic and the per-phase polynomial work follow SURVEY.md §3.1 steps 1-9 / App. C ([3P-MEM]: the
pinned crate is not on this machine).  Witness synthesis (Circuit::synthesize, :351-733) is host code the north star leaves
alone: this function receives its OUTPUT — the advice columns — exactly where halo2's prover has them after phase 2.
Every O(n) step runs on the GPU through the product API and stays resident in HBM between phases:
    commitments            zk_msm_batch_dev                 (one call per transcript phase)
    theta-compression      expression programs on the quotient interpreter
    permute_expression_pair / grand products               zk_lookup_permute_dev, zk_lookup_product_dev, zk_permutation_product_dev
    lagrange_to_coeff / coeff_to_extended / evaluate_h / divide_by_vanishing / extended_to_coeff
    evaluations            zk_eval_polynomial_batch_dev
    SHPLONK                zk_fr_lincomb_dev, zk_kate_division_dev
The host does what it does in the reference: Fiat-Shamir hashing, point encoding, O(#columns) bookkeeping.
Single circuit instance per proof (the reference passes `&[circuit]`); no user challenges / multi-phase advice.
"""
from __future__ import annotations

import threading

import time
from typing import List, Optional, Sequence

import numpy as np

from ..fields import R_MOD, fr_int, fr_mont, fr_mont_array, g1_affine_ints, omega, rand_fr_array
from ..kzg import ParamsKZG
from ..permutation import lookup_commit_products, permutation_commit, permute_expression_pairs
from .circuit import ADVICE, FIXED, INSTANCE
from .keygen import ProvingKey
from .shplonk import ProverQuery, ProverSHPLONK


def rotate_omega(x: int, rot: int, k: int) -> int:
    """EvaluationDomain::rotate_omega."""
    w = omega(k)
    return x * pow(w, rot % (1 << k), R_MOD) % R_MOD


def draw_plan(n_advice: int, n_lookups: int, n_sets: int, n_pieces: int, n: int, bf: int):
    """The order in which create_proof consumes the caller's `&mut rng`, as a list of (purpose, index, count of Fr::random draws, squeeze) items; `squeeze` names the
    challenge squeezed AFTER the item's phase (the item is drawn before that squeeze happens in halo2) — tests/test_rust_vectors.py compares the running totals
    with the Rust prover's counting RNG.  This is halo2_proofs v2023_01_20 (PSE; stack A of the reference) draw by draw ([3P-MEM]; include/zkmi355.h zk_rng_fn lists
    the source files): every commitment also draws one Blind(Fr::random) that KZG discards ("blind" items).  csrc/prover.hip builds the same plan (draw_schedule 1)."""
    plan = [("advice", i, bf + 1, "theta") for i in range(n_advice)]      # rows [usable_rows, n): the bf blinding rows and the one after
    plan += [("blind", None, 1, "theta")] * n_advice
    for l in range(n_lookups):
        plan += [("bi", l, bf + 1, "beta"), ("bt", l, bf + 1, "beta"), ("blind", None, 1, "beta"), ("blind", None, 1, "beta")]
    for s in range(n_sets):
        plan += [("perm_blind", s, bf, "y"), ("blind", None, 1, "y")]
    for l in range(n_lookups):
        plan += [("lookup_blind", l, bf, "y"), ("blind", None, 1, "y")]
    plan += [("random_poly", 0, n, "y"), ("blind", None, 1, "y")] + [("blind", None, 1, "x")] * n_pieces
    return plan


class _Joiner:
    """lets the helper thread sit in the `owned` list: create_proof's cleanup calls .free() on every entry"""

    def __init__(self, thread):
        self.thread = thread

    def free(self):
        self.thread.join()


def create_proof(params: ParamsKZG, pk: ProvingKey, advice: Sequence, instances: Sequence[Sequence[int]], rng: np.random.Generator, transcript,
                 timings: Optional[dict] = None, capture: Optional[dict] = None, phase_io=None) -> dict:
    """advice: cs.num_advice_columns columns of n rows — (n, 4) uint64 Montgomery host arrays or device buffers; rows past
    `usable_rows` are overwritten with blinding and device buffers are consumed (they hold coefficients afterwards).  instances:
    canonical ints per instance column.  Writes the proof into `transcript` and returns bookkeeping for tests / benches
    ({"commitments", "evals", "h_eval"}); `timings`, when given, receives wall milliseconds per phase.  Device buffers allocated
    along the way are released (back to the backend's pool) on success and on failure alike.  `capture` (measurement tooling only): receives host
    copies of the proof's committed Lagrange columns and challenges, so that a driver of the per-call host-buffer entry points can replay them."""
    owned: List = []
    try:
        return _create_proof(params, pk, advice, instances, rng, transcript, timings, owned, capture, phase_io)
    finally:
        for d in owned:
            d.free()


def _create_proof(params, pk, advice, instances, rng, transcript, timings, owned, capture=None, phase_io=None) -> dict:
    be, cs, k, n = pk.backend, pk.vk.cs, params.k, params.n
    dom = pk.domain
    ek, en = dom.extended_k, dom.extended_n
    bf = cs.blinding_factors()
    usable = n - (bf + 1)
    L = len(cs.lookups)
    assert len(advice) == cs.num_advice_columns and len(instances) == cs.num_instance_columns
    clock = [time.perf_counter()]

    def lap(name):
        if timings is not None:
            now = time.perf_counter()
            timings[name] = timings.get(name, 0.0) + (now - clock[0]) * 1e3
            clock[0] = now

    def dev(nbytes):
        d = be.alloc(nbytes)
        owned.append(d)
        return d

    # `phase_io` (measurement tooling only, bench.py extra.phase_batched_shim): the integration level between the thin shim and the resident prover — halo2's
    # Polynomial values stay HOST vectors and every phase is a batch of device calls.  phase_io.reads(phase, columns) runs before a phase touches witness-dependent
    # columns it did not produce itself (a host-resident prover uploads them), phase_io.writes(phase, columns) after it produced columns halo2 keeps (it downloads
    # them).  The hooks move the bytes for real and change no value: the proof is the same.
    def reads(phase, cols):
        if phase_io is not None:
            phase_io.reads(phase, [c for c in cols if c is not None])

    def writes(phase, cols):
        if phase_io is not None:
            phase_io.writes(phase, [c for c in cols if c is not None])

    def commit_all(which, cols):                                 # one device call per transcript phase (all-gathered when the tables are sharded)
        return [g1_affine_ints(r) for r in params.commit_columns(which, cols)]

    # ---- 1. vk, instances ---------------------------------------------------------------------------------------------
    pk.vk.hash_into(transcript)
    inst_values = []
    for col in instances:
        assert len(col) <= usable, "instance column longer than the usable rows"
        for v in col:                                             # KZG: QUERY_INSTANCE = false -> values go straight into the transcript
            transcript.common_scalar(v)
        a = np.zeros((n, 4), dtype=np.uint64)
        if len(col):
            a[: len(col)] = fr_mont_array(col)
        d = dev(n * 32)
        d.upload(a)
        inst_values.append(d)

    lap("1_instances")
    # ---- 2. advice: blind the unusable rows, commit ----------------------------------------------------------------------
    # host columns (what halo2's prover holds after synthesis: Vec<Fr> per column) cross PCIe here, all in one call; page-locked arrays
    # (Backend.host_alloc) travel at link rate.  Device buffers are taken as they are.
    adv_values = [dev(n * 32) if isinstance(col, np.ndarray) else col for col in advice]
    from_host = [i for i, col in enumerate(advice) if isinstance(col, np.ndarray)]
    if from_host:
        be.upload_columns([adv_values[i] for i in from_host], [np.ascontiguousarray(advice[i], dtype=np.uint64).reshape(n, 4) for i in from_host], n * 32)
    # Every `Fr::random` draw of the proof (blinding rows, the n coefficients of the vanishing argument's random polynomial, the Blind every commitment
    # draws and KZG discards) depends on no challenge: the advice rows are drawn here, a helper thread draws the rest, in draw_plan's order (the stream —
    # hence the proof — is the same), while the GPU commits the advice columns; a phase waits only for its own items.
    chunk = cs.permutation_chunk_len()
    n_sets = (len(cs.permutation_columns) + chunk - 1) // chunk if cs.permutation_columns else 0
    plan = draw_plan(len(advice), L, n_sets, dom.quotient_poly_degree, n, bf)
    blinds = [rand_fr_array(rng, cnt) for _, _, cnt, _ in plan[:len(advice)]]
    if advice:
        be.upload_columns([d.ptr + usable * 32 for d in adv_values], blinds, (n - usable) * 32)
    drawn = {name: {} for name in ("bi", "bt", "perm_blind", "lookup_blind", "random_poly")}
    ready = {name: threading.Event() for name in drawn}
    want = {name: sum(1 for it_ in plan if it_[0] == name) for name in drawn}
    for name, cnt in want.items():
        if cnt == 0:
            ready[name].set()
    draw_error = []

    def draw_all():
        try:
            for name, idx, cnt, _ in plan[len(advice):]:
                a = rand_fr_array(rng, cnt)
                if name == "blind":
                    continue
                drawn[name][idx] = a
                if len(drawn[name]) == want[name]:
                    ready[name].set()
        except BaseException as e:                                   # never leave the main thread waiting
            draw_error.append(e)
        finally:
            for ev_ in ready.values():
                ev_.set()

    def take_draw(name):
        ready[name].wait()
        if draw_error:
            raise draw_error[0]
        items = [drawn[name][i] for i in range(want[name])]
        if name == "random_poly":
            return items[0]
        if name == "perm_blind":
            return items
        return np.stack(items) if items else np.zeros((0, bf + 1 if name in ("bi", "bt") else bf, 4), np.uint64)
    drawer = threading.Thread(target=draw_all)
    drawer.start()
    owned.append(_Joiner(drawer))                                    # joined on every exit path (the generator belongs to the caller again afterwards)
    for pt in commit_all("g_lagrange", adv_values):
        transcript.write_point(pt)

    lap("2_advice_commit")
    # ---- 3. theta; lookups: compress, permute, commit --------------------------------------------------------------------
    theta = transcript.squeeze_challenge()
    one = fr_mont(1)
    th = fr_mont(theta)
    reads("3_lookup_permuted", adv_values + inst_values)
    compressed = []
    table_cache = {}                                            # lookups with the same table expressions share ONE compressed table column
    anycol = pk.fixed_values[0] if pk.fixed_values else adv_values[0]

    def run_compressor(evl):
        out = dev(n * 32)
        evl.evaluate_h(fixed=pk.fixed_values, advice=adv_values, instance=inst_values, l0=anycol, l_last=anycol, l_active_row=anycol,
                       perm_cosets=[], perm_products=[], lookup_product=[], lookup_input=[], lookup_table=[], challenges=[],
                       beta=one, gamma=one, theta=th, y=one, out=out)
        return out
    for lk, (cin_ev, ctab_ev) in zip(cs.lookups, pk.lookup_compressors):
        key = tuple(lk.table_expressions)
        if key not in table_cache:
            table_cache[key] = run_compressor(ctab_ev)
        compressed.append([run_compressor(cin_ev), table_cache[key]])
    # permute_expression_pair of every lookup in one device call (raises ZkError when an input is not in the table)
    bi, bt = take_draw("bi"), take_draw("bt")
    permuted = permute_expression_pairs([c[0] for c in compressed], [c[1] for c in compressed], k, bf, bi, bt, backend=be)
    for a_, s_ in permuted:
        owned += [a_, s_]
    flat = [c for pr in permuted for c in pr]
    writes("3_lookup_permuted", flat + [c[0] for c in compressed] + list(table_cache.values()))
    for pt in commit_all("g_lagrange", flat):           # per lookup: permuted input, permuted table
        transcript.write_point(pt)

    lap("3_lookup_permuted")
    # ---- 4. beta, gamma; permutation and lookup grand products ---------------------------------------------------------------
    beta = transcript.squeeze_challenge()
    gamma = transcript.squeeze_challenge()
    bt_m, gm_m = fr_mont(beta), fr_mont(gamma)
    perm_values = []
    for t, i in cs.permutation_columns:
        perm_values.append({ADVICE: adv_values, FIXED: pk.fixed_values, INSTANCE: inst_values}[t][i])
    perm_blind, lookup_blind = take_draw("perm_blind"), take_draw("lookup_blind")
    reads("4_grand_products", [c for c in perm_values if c in adv_values or c in inst_values] + flat + [c[0] for c in compressed] + list(table_cache.values()))
    zs = permutation_commit(perm_values, pk.sigma_values, k, cs.degree(), bt_m, gm_m, perm_blind, backend=be) if perm_values else []
    owned += zs
    lzs = lookup_commit_products([(c[0], c[1], p_[0], p_[1]) for c, p_ in zip(compressed, permuted)], k, bt_m, gm_m, lookup_blind, backend=be)
    owned += lzs
    writes("4_grand_products", zs + lzs)
    for pt in commit_all("g_lagrange", zs + lzs):      # permutation products, then lookup products: one MSM batch, transcript order kept
        transcript.write_point(pt)

    lap("4_grand_products")
    # ---- 5. vanishing argument: random polynomial -----------------------------------------------------------------------------
    random_poly = dev(n * 32)
    random_poly.upload(take_draw("random_poly"))
    transcript.write_point(commit_all("g", [random_poly])[0])

    lap("5_random_poly")
    # ---- 6. y; everything to coefficient form; h(X) ------------------------------------------------------------------------------
    y = transcript.squeeze_challenge()
    lag = adv_values + inst_values + zs + lzs + flat                 # the witness-dependent columns, Lagrange basis
    if capture is not None:
        capture.update(advice=[d.download((n, 4)) for d in adv_values], perm_products=[d.download((n, 4)) for d in zs],
                       lookup_products=[d.download((n, 4)) for d in lzs], permuted=[(a_.download((n, 4)), s_.download((n, 4))) for a_, s_ in permuted],
                       compressed=[(c[0].download((n, 4)), c[1].download((n, 4))) for c in compressed], random_poly=drawn["random_poly"][0],
                       theta=theta, beta=beta, gamma=gamma, y=y)
    reads("6_ntt_evaluate_h", lag)
    be.lagrange_to_coeff_batch_dev(lag, k)                           # in place: from here on these buffers hold coefficients
    writes("6_ntt_evaluate_h", lag)                                  # (halo2 keeps advice_polys / permuted polys / product polys for the evaluation phase)
    adv_polys, inst_polys = adv_values, inst_values
    nA, nI, nZ = len(adv_polys), len(inst_polys), len(zs)

    def split(ext):
        return dict(advice=ext[:nA], instance=ext[nA:nA + nI], perm_products=ext[nA + nI:nA + nI + nZ], lookup_product=ext[nA + nI + nZ:nA + nI + nZ + L],
                    lookup_input=ext[nA + nI + nZ + L:][0::2], lookup_table=ext[nA + nI + nZ + L:][1::2])
    scal = dict(challenges=[], beta=bt_m, gamma=gm_m, theta=th, y=fr_mont(y))
    n_pieces = dom.quotient_poly_degree
    h_ext = dev(en * 32) if not pk.pieces_from_cosets else None
    numer = []
    if pk.pieces_from_cosets:
        # cs_degree - 1 cosets determine h(X) (deg h < (cs_degree - 1) n; on a coset X^n is a constant): evaluate the numerator there only, the pieces come from zk_cosets_to_pieces_dev
        ext = [dev(n * 32) for _ in lag]
        numer = [dev(n * 32) for _ in range(n_pieces)]
        for j in range(n_pieces):
            part = pk.coset_parts[j]
            be.coeff_to_coset_batch_dev(lag, ext, k, ek, j)
            pk.evaluator.evaluate_h(fixed=part["fixed"], l0=part["l"][0], l_last=part["l"][1], l_active_row=part["l"][2], perm_cosets=part["sigma"],
                                    out=numer[j], coset=j, **split(ext), **scal)
    elif pk.coset_parts is None:
        ext = [dev(en * 32) for _ in lag]
        be.coeff_to_extended_batch_dev(lag, ext, k, ek)
        pk.evaluator.evaluate_h(fixed=pk.fixed_cosets, l0=pk.l0, l_last=pk.l_last, l_active_row=pk.l_active_row, perm_cosets=pk.sigma_cosets,
                                out=h_ext, **split(ext), **scal)
    else:
        # the quotient sharded by coset: this rank brings the columns to ITS cosets only (size-n NTTs), evaluates the numerator
        # there, and the ranks exchange the numerator values — the one bulk collective of a proof (n * 32 bytes per coset)
        # (with more ranks than cosets the ranks of a coset split its ROWS: each still runs that coset's NTTs, and evaluates its slice)
        n_cosets = 1 << (ek - k)
        parts = params.quotient_parts(n_cosets)
        rows = n // parts
        slots = -(-(n_cosets * parts) // params.world)
        xch = params.coset_exchange(slots * rows * 32) if params.coset_exchange is not None else None
        mine = np.zeros((slots, rows, 4), dtype=np.uint64) if xch is None else None
        ext = [dev(n * 32) for _ in range(max(len(lag) + 1, n_cosets))]
        cols, num = ext[:len(lag)], ext[len(lag)]
        at = None
        for s_, (j, lo, cnt) in enumerate(params.my_units(n_cosets)):
            part = pk.coset_parts[j]
            if j != at:
                be.coeff_to_coset_batch_dev(lag, cols, k, ek, j)
                at = j
            pk.evaluator.evaluate_h(fixed=part["fixed"], l0=part["l"][0], l_last=part["l"][1], l_active_row=part["l"][2], perm_cosets=part["sigma"],
                                    out=num if xch is None else xch[0] + s_ * rows * 32, coset=j, rows=None if parts == 1 else (lo, cnt), **split(cols), **scal)
            if xch is None:
                mine[s_] = num.download((rows, 4))
        if xch is None:
            every = params.gather_cosets(mine, n_cosets)
            for j in range(n_cosets):
                ext[j].upload(np.ascontiguousarray(every[j]))
            srcs = ext[:n_cosets]
        else:                                                       # device-resident exchange (RCCL all_gather on the caller's buffers)
            xch[2]()
            srcs = [xch[1] + j * n * 32 for j in range(n_cosets)]   # rank r's block starts at r * slots * rows * 32: unit order = coset order, rows ascending
        be.fr_interleave_dev(srcs, n, h_ext)
    for d in ext:                                                   # the cosets are dead once the numerator exists
        d.free()
        owned.remove(d)
    lap("6_ntt_evaluate_h")
    # ---- 7. vanishing::construct: divide, back to coefficients, split into d-1 pieces, commit ------------------------------------------
    if pk.pieces_from_cosets:
        piece_bufs = [dev(n * 32) for _ in range(n_pieces)]
        be.cosets_to_pieces_dev(numer, k, ek, piece_bufs)
        pieces = [b.ptr for b in piece_bufs]
    else:
        be.divide_by_vanishing_poly_dev(h_ext, k, ek)
        be.extended_to_coeff_dev(h_ext, k, ek)
        pieces = [h_ext.ptr + i * n * 32 for i in range(n_pieces)]
    writes("7_h_construct_commit", pieces)
    for pt in commit_all("g", pieces):
        transcript.write_point(pt)

    lap("7_h_construct_commit")
    # ---- 8. x; evaluations ---------------------------------------------------------------------------------------------------------------
    x = transcript.squeeze_challenge()
    xn = pow(x, n, R_MOD)
    rot = lambda r: rotate_omega(x, r, k)
    # h(X) as ONE polynomial of n coefficients: sum_i xn^i piece_i(X) (vanishing::Constructed::evaluate)
    reads("8_evaluations", lag + [random_poly] + pieces)
    h_poly = dev(n * 32)
    be.fr_lincomb_dev(pieces, fr_mont_array([pow(xn, i, R_MOD) for i in range(n_pieces)]), n, h_poly)
    x_last = rot(-(bf + 1))
    ev_polys, ev_points = [], []

    def ask(poly, point):
        ev_polys.append(poly)
        ev_points.append(point)
    for c, r in cs.advice_queries():
        ask(adv_polys[c], rot(r))
    for c, r in cs.fixed_queries():
        ask(pk.fixed_polys[c], rot(r))
    ask(random_poly, x)
    for s in pk.sigma_polys:                                         # permutation "common" evaluations
        ask(s, x)
    for i, z in enumerate(zs):
        ask(z, x)
        ask(z, rot(1))
        if i + 1 < len(zs):
            ask(z, x_last)
    for z, (a, s) in zip(lzs, permuted):
        ask(z, x)
        ask(z, rot(1))
        ask(a, x)
        ask(a, rot(-1))
        ask(s, x)
    ask(h_poly, x)                                                  # not written to the proof: the verifier derives it
    evals_m = be.eval_polynomial_batch_dev(ev_polys, n, fr_mont_array(ev_points))
    evals = [fr_int(e) for e in evals_m]
    for e in evals[:-1]:
        transcript.write_scalar(e)

    lap("8_evaluations")
    # ---- 9. multi-open ------------------------------------------------------------------------------------------------------------------
    it = iter(zip(ev_polys, ev_points, evals))
    take = lambda: ProverQuery(*next(it))
    q_adv = [take() for _ in cs.advice_queries()]
    q_fix = [take() for _ in cs.fixed_queries()]
    q_rand = take()
    q_sigma = [take() for _ in pk.sigma_polys]
    q_perm_a, q_perm_last = [], []
    for i in range(len(zs)):
        q_perm_a += [take(), take()]
        if i + 1 < len(zs):
            q_perm_last.append(take())
    q_lk = []
    for _ in lzs:
        pz, pzn, pa, pai, ps = take(), take(), take(), take(), take()
        q_lk += [pz, pa, ps, pai, pzn]                              # lookup::Evaluated::open order
    q_h = take()
    reads("9_shplonk", lag + [random_poly, h_poly])
    queries = q_adv + q_perm_a + list(reversed(q_perm_last)) + q_lk + q_fix + q_sigma + [q_h, q_rand]
    ProverSHPLONK(params).create_proof(transcript, queries)
    lap("9_shplonk")
    return {"commitments": nA + 2 * L + nZ + L + 1 + n_pieces + 2, "evals": len(evals) - 1, "h_eval": evals[-1]}
