"""Synthetic proving-key programs for the evaluate_h parity tests (the real circuit's constraint
system cannot be extracted without the Rust toolchain; SURVEY.md §3.1 gives its shape)."""
import random

import numpy as np

from zk_dcap_verifier_amd import evaluation as ev


def build_program(orc, pyref, k, cs_degree, n_fixed, n_advice, n_instance, n_challenges, n_perm, n_lookups, seed, gate_ops=24):
    """Random but well-formed Program exercising every Calculation / ValueSource variant."""
    rnd = random.Random(seed)
    ek = k
    while (1 << ek) < (1 << k) * (cs_degree - 1):
        ek += 1

    def rand_graph(n_ops, with_prev):
        g = ev.Graph()
        consts = [g.add_constant(orc.fr_from_ints([rnd.randrange(pyref.R)])[0]) for _ in range(3)]
        vals = list(consts)

        def leaf():
            kinds = []
            if n_fixed:
                kinds.append("f")
            if n_advice:
                kinds += ["a", "a"]
            if n_instance:
                kinds.append("i")
            if n_challenges:
                kinds.append("c")
            kinds += ["beta", "gamma", "theta", "y"]
            if with_prev:
                kinds.append("prev")
            t = rnd.choice(kinds)
            rot = g.add_rotation(rnd.choice([0, 0, 1, -1, 2, -3]))
            return {"f": lambda: ev.vs(ev.FIXED, rnd.randrange(n_fixed), rot), "a": lambda: ev.vs(ev.ADVICE, rnd.randrange(n_advice), rot),
                    "i": lambda: ev.vs(ev.INSTANCE, rnd.randrange(n_instance), rot), "c": lambda: ev.vs(ev.CHALLENGE, rnd.randrange(n_challenges)),
                    "beta": lambda: ev.vs(ev.BETA), "gamma": lambda: ev.vs(ev.GAMMA), "theta": lambda: ev.vs(ev.THETA), "y": lambda: ev.vs(ev.Y),
                    "prev": lambda: ev.vs(ev.PREVIOUS)}[t]()

        def operand():
            return rnd.choice(vals) if vals and rnd.random() < 0.5 else leaf()

        for _ in range(n_ops):
            op = rnd.choice([ev.ADD, ev.SUB, ev.MUL, ev.MUL, ev.SQUARE, ev.DOUBLE, ev.NEGATE, ev.HORNER, ev.STORE])
            if op in (ev.ADD, ev.SUB, ev.MUL):
                v = g.add_calculation(op, operand(), operand())
            elif op == ev.HORNER:
                v = g.add_calculation(op, operand(), [operand() for _ in range(rnd.randrange(0, 4))], operand())
            else:
                v = g.add_calculation(op, operand())
            vals.append(v)
        return g

    custom = rand_graph(gate_ops, True)
    # halo2's Evaluator::new ends the custom-gate graph with Horner(PreviousValue, gate polys, Y)
    parts = [ev.vs(ev.INTERMEDIATE, i) for i in range(custom.num_intermediates)][-4:]
    custom.add_calculation(ev.HORNER, ev.vs(ev.PREVIOUS), parts, ev.vs(ev.Y))
    lookups = []
    for _ in range(n_lookups):
        g = rand_graph(8, False)
        # (compressed_input + beta) * (compressed_table + gamma) shape as the last calculation
        a = g.add_calculation(ev.ADD, ev.vs(ev.INTERMEDIATE, g.num_intermediates - 1), ev.vs(ev.BETA))
        b = g.add_calculation(ev.ADD, ev.vs(ev.INTERMEDIATE, g.num_intermediates - 3), ev.vs(ev.GAMMA))
        g.add_calculation(ev.MUL, a, b)
        lookups.append(g)
    perm_columns = []
    pools = [(0, n_advice), (1, n_fixed), (2, n_instance)]
    for _ in range(n_perm):
        t, cnt = rnd.choice([p for p in pools if p[1] > 0])
        perm_columns.append((t, rnd.randrange(cnt)))
    return ev.Program(k=k, extended_k=ek, n_fixed=n_fixed, n_advice=n_advice, n_instance=n_instance, n_challenges=n_challenges,
                      blinding_factors=5, cs_degree=cs_degree, perm_columns=perm_columns, custom_gates=custom, lookups=lookups)


def rand_cols(pc_mod, orc, pyref, count, size, seed):
    return [pc_mod.rand_fr(orc, pyref, size, seed + 7 * i) for i in range(count)]


def run_case(be, orc, pyref, pc_mod, prog, seed=5, check_cosets=True, expect_kernels=False):
    size = 1 << prog.extended_k
    chunk = prog.cs_degree - 2
    n_sets = (len(prog.perm_columns) + chunk - 1) // chunk if prog.perm_columns else 0
    nl = len(prog.lookups)
    cols = dict(fixed=rand_cols(pc_mod, orc, pyref, prog.n_fixed, size, seed), advice=rand_cols(pc_mod, orc, pyref, prog.n_advice, size, seed + 100),
                instance=rand_cols(pc_mod, orc, pyref, prog.n_instance, size, seed + 200),
                perm_cosets=rand_cols(pc_mod, orc, pyref, len(prog.perm_columns), size, seed + 300),
                perm_products=rand_cols(pc_mod, orc, pyref, n_sets, size, seed + 400),
                lookup_product=rand_cols(pc_mod, orc, pyref, nl, size, seed + 500), lookup_input=rand_cols(pc_mod, orc, pyref, nl, size, seed + 600),
                lookup_table=rand_cols(pc_mod, orc, pyref, nl, size, seed + 700))
    l0, l_last, l_active = rand_cols(pc_mod, orc, pyref, 3, size, seed + 800)
    chal = pc_mod.rand_fr(orc, pyref, max(prog.n_challenges, 1), seed + 900)[: prog.n_challenges]
    beta, gamma, theta, y = pc_mod.rand_fr(orc, pyref, 4, seed + 901)
    want = orc.evaluate_h(prog.to_blob(), cols["fixed"], cols["advice"], cols["instance"], l0, l_last, l_active, cols["perm_cosets"],
                          cols["perm_products"], cols["lookup_product"], cols["lookup_input"], cols["lookup_table"], chal, beta, gamma, theta, y, size)
    dev = {k: [be.to_device(c) for c in v] for k, v in cols.items()}
    d_l0, d_ll, d_la = be.to_device(l0), be.to_device(l_last), be.to_device(l_active)
    out = be.alloc(size * 32)
    e = ev.Evaluator(prog, backend=be)
    if expect_kernels:                                                 # tests/test_quotient_jit.py: the program must run on kernels generated for it, not on the interpreter
        assert be.quotient_program_kernels(e.handle) >= (2 if expect_kernels is True else int(expect_kernels)), "the program was loaded without generated kernels (tune quot_jit)"
    e.evaluate_h(fixed=dev["fixed"], advice=dev["advice"], instance=dev["instance"], l0=d_l0, l_last=d_ll, l_active_row=d_la,
                 perm_cosets=dev["perm_cosets"], perm_products=dev["perm_products"], lookup_product=dev["lookup_product"],
                 lookup_input=dev["lookup_input"], lookup_table=dev["lookup_table"], challenges=chal, beta=beta, gamma=gamma, theta=theta, y=y, out=out)
    got = out.download((size, 4))
    # the same numerator coset by coset (zk_quotient_run_coset_dev: columns = every 2^e-th row, rotations step by one row) and on aligned slices of a
    # coset's rows (zk_quotient_run_coset_rows_dev) — the units of the multi-GPU quotient
    e_bits = prog.extended_k - prog.k
    if check_cosets and e_bits >= 1:
        n, nc = 1 << prog.k, 1 << e_bits
        for j in ([0, nc - 1] if nc > 2 else range(nc)):
            cdev = {k_: [be.to_device(np.ascontiguousarray(c[j::nc])) for c in v] for k_, v in cols.items()}
            cl = [be.to_device(np.ascontiguousarray(c[j::nc])) for c in (l0, l_last, l_active)]
            co = be.alloc(n * 32)
            kw = dict(fixed=cdev["fixed"], advice=cdev["advice"], instance=cdev["instance"], l0=cl[0], l_last=cl[1], l_active_row=cl[2],
                      perm_cosets=cdev["perm_cosets"], perm_products=cdev["perm_products"], lookup_product=cdev["lookup_product"],
                      lookup_input=cdev["lookup_input"], lookup_table=cdev["lookup_table"], challenges=chal, beta=beta, gamma=gamma, theta=theta, y=y)
            e.evaluate_h(out=co, coset=j, **kw)
            assert (co.download((n, 4)) == want[j::nc]).all(), ("coset", j)
            if be.quotient_program_split(e.handle)["low_cosets"]:      # the two parts on coset-layout columns (zk_quotient_run_coset_part_dev) add up to the coset's numerator
                p1, p2 = be.alloc(n * 32), be.alloc(n * 32)
                e.evaluate_h(out=p1, coset=j, part=1, **kw)
                e.evaluate_h(out=p2, coset=j, part=2, **kw)
                assert (orc.fr_add(p1.download((n, 4)), p2.download((n, 4))) == want[j::nc]).all(), ("split, coset layout", j)
                p1.free(); p2.free()
            parts = 4 if n >= 4 else 1
            rows = n // parts
            for p_ in range(parts):
                so = be.alloc(rows * 32)
                e.evaluate_h(out=so, coset=j, rows=(p_ * rows, rows), **kw)
                assert (so.download((rows, 4)) == want[j::nc][p_ * rows:(p_ + 1) * rows]).all(), ("coset rows", j, p_)
                so.free()
            for v in cdev.values():
                for d in v:
                    d.free()
            for d in cl + [co]:
                d.free()
    # the degree split (include/zkmi355.h): on every row the numerator is the sum of its high and low parts — for ANY column values, the split is pure algebra
    # (a sum of y-weighted identities regrouped) — with the low part evaluated on the rows of cosets 0 and 1 only, coset-major
    sp = be.quotient_program_split(e.handle)
    if sp["low_cosets"]:
        n, nc, lc = 1 << prog.k, 1 << e_bits, sp["low_cosets"]
        assert lc == 2 and sp["instructions_high"] and sp["instructions_low"]
        kw = dict(fixed=dev["fixed"], advice=dev["advice"], instance=dev["instance"], l0=d_l0, l_last=d_ll, l_active_row=d_la, perm_cosets=dev["perm_cosets"],
                  perm_products=dev["perm_products"], lookup_product=dev["lookup_product"], lookup_input=dev["lookup_input"], lookup_table=dev["lookup_table"],
                  challenges=chal, beta=beta, gamma=gamma, theta=theta, y=y)
        hi, lo = be.alloc(size * 32), be.alloc(lc * n * 32)
        e.evaluate_h(out=hi, part=1, **kw)
        e.evaluate_h(out=lo, part=2, low_cosets=lc, **kw)
        h_hi, h_lo = hi.download((size, 4)), lo.download((lc * n, 4))
        for j in range(lc):
            assert (orc.fr_add(np.ascontiguousarray(h_hi[j::nc]), np.ascontiguousarray(h_lo[j * n:(j + 1) * n])) == want[j::nc]).all(), ("split, extended layout, coset", j)
        hi.free(); lo.free()
    e.release()
    for v in dev.values():
        for d in v:
            d.free()
    for d in (d_l0, d_ll, d_la, out):
        d.free()
    assert (got == want).all(), np.nonzero((got != want).any(axis=1))[0][:8]
    return got
