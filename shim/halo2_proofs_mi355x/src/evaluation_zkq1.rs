// include!("evaluation_zkq1.rs") at the end of halo2_proofs/src/plonk/evaluation.rs (zkwebauthn/halo2 @ c254c75): the types below are private to that
// module, which is why this is an include and not a module.  [3P-MEM]: field and variant names follow PSE halo2 v2023_01_20's evaluation.rs
// (GraphEvaluator { constants, rotations, calculations, num_intermediates }, CalculationInfo { calculation, target }, Calculation::{Add, Sub, Mul, Square,
// Double, Negate, Horner, Store}, ValueSource::{Constant, Intermediate, Fixed, Advice, Instance, Challenge, Beta, Gamma, Theta, Y, PreviousValue});
// check them against the checkout before building.  Format = INTEGRATION.md §3 (little-endian u32 words; Fr constants as 8 words, Montgomery limbs as in
// memory); zk-dcap-verifier_amd/evaluation.py `Program.to_blob()` is the Python twin and oracle/evaluate_h_oracle.inc an independent reader.

const ZKQ1_MAGIC: u32 = 0x3151_4B5A;

fn zkq1_value_source(out: &mut Vec<u32>, v: &ValueSource) {
    let (kind, a, b) = match *v {
        ValueSource::Constant(i) => (0, i, 0),
        ValueSource::Intermediate(i) => (1, i, 0),
        ValueSource::Fixed(col, rot) => (2, col, rot),
        ValueSource::Advice(col, rot) => (3, col, rot),
        ValueSource::Instance(col, rot) => (4, col, rot),
        ValueSource::Challenge(i) => (5, i, 0),
        ValueSource::Beta() => (6, 0, 0),
        ValueSource::Gamma() => (7, 0, 0),
        ValueSource::Theta() => (8, 0, 0),
        ValueSource::Y() => (9, 0, 0),
        ValueSource::PreviousValue() => (10, 0, 0),
    };
    out.extend_from_slice(&[kind as u32, a as u32, b as u32]);
}

fn zkq1_graph<C: CurveAffine>(out: &mut Vec<u32>, g: &GraphEvaluator<C>) {
    out.push(g.constants.len() as u32);
    for c in &g.constants {
        // the in-memory Montgomery limbs of bn256::Fr ([u64; 4]), exactly what the kernels compute on
        assert_eq!(std::mem::size_of::<C::ScalarExt>(), 32);
        let limbs: [u32; 8] = unsafe { std::mem::transmute_copy(c) };
        out.extend_from_slice(&limbs);
    }
    out.push(g.rotations.len() as u32);
    out.extend(g.rotations.iter().map(|r| *r as u32)); // i32 two's complement
    out.push(g.num_intermediates as u32);
    out.push(g.calculations.len() as u32);
    for info in &g.calculations {
        let (op, unary): (u32, Option<&ValueSource>) = match &info.calculation {
            Calculation::Add(..) => (0, None),
            Calculation::Sub(..) => (1, None),
            Calculation::Mul(..) => (2, None),
            Calculation::Square(a) => (3, Some(a)),
            Calculation::Double(a) => (4, Some(a)),
            Calculation::Negate(a) => (5, Some(a)),
            Calculation::Horner(..) => (6, None),
            Calculation::Store(a) => (7, Some(a)),
        };
        out.push(op);
        out.push(info.target as u32);
        match &info.calculation {
            Calculation::Add(a, b) | Calculation::Sub(a, b) | Calculation::Mul(a, b) => {
                zkq1_value_source(out, a);
                zkq1_value_source(out, b);
            }
            Calculation::Horner(start, parts, factor) => {
                zkq1_value_source(out, start);
                zkq1_value_source(out, factor);
                out.push(parts.len() as u32);
                for p in parts {
                    zkq1_value_source(out, p);
                }
            }
            _ => zkq1_value_source(out, unary.unwrap()),
        }
    }
}

impl<C: CurveAffine> Evaluator<C> {
    /// The whole compiled evaluator of a proving key as one ZKQ1 blob (input of zk_quotient_program_load).
    pub(crate) fn to_zkq1(&self, cs: &ConstraintSystem<C::ScalarExt>, k: u32, extended_k: u32) -> Vec<u8> {
        let mut w: Vec<u32> = vec![
            ZKQ1_MAGIC,
            k,
            extended_k,
            cs.num_fixed_columns as u32,
            cs.num_advice_columns as u32,
            cs.num_instance_columns as u32,
            cs.num_challenges as u32,
            cs.blinding_factors() as u32,
            cs.degree() as u32,
        ];
        let cols = cs.permutation.get_columns();
        w.push(cols.len() as u32);
        for c in &cols {
            let ty = match c.column_type() {
                Any::Advice(_) => 0u32, // (Any::Advice without payload in forks that predate multi-phase advice)
                Any::Fixed => 1,
                Any::Instance => 2,
            };
            w.extend_from_slice(&[ty, c.index() as u32]);
        }
        w.push(self.lookups.len() as u32);
        zkq1_graph(&mut w, &self.custom_gates);
        for g in &self.lookups {
            zkq1_graph(&mut w, g);
        }
        w.iter().flat_map(|x| x.to_le_bytes()).collect()
    }
}

/// `compress_expressions` of lookup::Argument::commit_permuted (theta-compression of a lookup's input or table expressions over the n = 2^k rows) as a ZKQ1
/// program: header with extended_k = k, no permutation columns, no lookups; the "custom gates" graph is `Horner(Constant(0), [e_0 .. e_m-1], Theta)`, the same
/// calculation Evaluator::new builds for the lookup's coset form.  Python twin: zk-dcap-verifier_amd/plonk/keygen.py `_compressor` + evaluation.py
/// `expression_program`.  One program per lookup side goes into ZkPlonkPkHost.lookup_{input,table}_zkq1 (pk_desc.rs).
pub(crate) fn lookup_expression_zkq1<C: CurveAffine>(cs: &ConstraintSystem<C::ScalarExt>, k: u32, exprs: &[Expression<C::ScalarExt>]) -> Vec<u8> {
    let mut graph = GraphEvaluator::<C>::default(); // constants [0, 1, 2]
    let parts: Vec<ValueSource> = exprs.iter().map(|e| graph.add_expression(e)).collect();
    graph.add_calculation(Calculation::Horner(ValueSource::Constant(0), parts, ValueSource::Theta()));
    let mut w: Vec<u32> = vec![
        ZKQ1_MAGIC,
        k,
        k, // extended_k = k: rotations wrap modulo n, as Expression::evaluate over Lagrange columns does
        cs.num_fixed_columns as u32,
        cs.num_advice_columns as u32,
        cs.num_instance_columns as u32,
        cs.num_challenges as u32,
        0, // blinding factors: unused by an expression program
        3, // cs_degree: the smallest the loader accepts
        0, // no permutation columns
        0, // no lookups
    ];
    zkq1_graph(&mut w, &graph);
    w.iter().flat_map(|x| x.to_le_bytes()).collect()
}

/// evaluate_h redirect: call at the top of Evaluator::evaluate_h; `None` = run the original body.  The program / proving-key handles are created
/// on first use and cached per ProvingKey address (keygen_pk's output lives as long as the prover uses it).
///   zk_pk_load(prog, pk.fixed_cosets, pk.permutation.cosets, pk.l0, pk.l_last, pk.l_active_row, form = 1)     — the cosets pk already stores
///   zk_evaluate_h(pk, advice_polys, instance_polys, permutation z polys, lookup {product, permuted_input, permuted_table} polys,
///                 challenges, beta, gamma, theta, y, finish = 0, out)                                          — coefficient form in, extended numerator out
/// With finish = 1 (and vanishing::Argument::construct skipping its divide_by_vanishing_poly + extended_to_coeff) h(X)'s (d-1)·n coefficients come back
/// instead; that edit is in prover_phases.patch.
pub(crate) fn mi355x_evaluate_h<C: CurveAffine>(/* same arguments as evaluate_h */) -> Option<Polynomial<C::ScalarExt, ExtendedLagrangeCoeff>> {
    // body: crate::mi355x::gpu()? ; per-pk cache lookup ; pointer arrays over the argument slices ; one zk_evaluate_h call ; wrap `out`.
    // Left as a comment on purpose: it is 30 lines of pointer plumbing whose exact argument types (Polynomial<_, Coeff> vs the
    // Committed / permutation::prover::Committed wrappers) must be read off the checkout; the C side is exercised by
    // tests/test_quotient.py::test_pk_level_evaluate_h and by bench.py's `extra.thin_shim`.
    None
}
