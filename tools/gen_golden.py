#!/usr/bin/env python3
"""Regenerate tests/golden/kats.json from the pure-integer definitions in oracle/pyref.py.

These are mathematical facts (SURVEY.md App. A [COMPUTED]), not outputs of the reference: the reference holds no
MSM / NTT vectors (parity unpinned).  The only reference-held datum under tests/golden/ is proof.bin, a byte-for-byte
copy of /root/reference/bin/assets/proof.bin (the data file of bin/src/main.rs:269-279)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyref as p  # noqa: E402


def main():
    G = p.G1_GEN
    kats = {
        "fq_modulus": hex(p.P), "fr_modulus": hex(p.R),
        "fq_R": hex(p.mont_r(p.P)), "fr_R": hex(p.mont_r(p.R)), "fr_R2": hex(p.mont_r(p.R) ** 2 % p.R),
        "fq_inv64": hex(p.mont_inv64(p.P)), "fr_inv64": hex(p.mont_inv64(p.R)),
        "root_of_unity": hex(p.ROOT_OF_UNITY), "delta": hex(p.DELTA), "zeta": hex(p.ZETA),
        "omega": {str(k): hex(p.omega(k)) for k in (2, 3, 17, 18, 19, 21, 22, 24)},
        "g1_multiples": {str(k): [hex(c) for c in p.g1_mul(G, k)] for k in (2, 3, 30, 52480)},
        "msm": [{"scalars": [1, 2, 3, 4], "base_dlogs": [1, 2, 3, 4], "result_dlog": 30},
                {"scalars": [i * i + 7 for i in range(16)], "base_dlogs": [5 + 3 * i for i in range(16)], "result_dlog": 52480}],
        "ntt": [{"log_n": 2, "input": [1, 2, 3, 4], "output": [hex(v) for v in p.ntt_definition([1, 2, 3, 4], p.omega(2))]},
                {"log_n": 3, "input": list(range(1, 9)), "output": [hex(v) for v in p.ntt_definition(list(range(1, 9)), p.omega(3))]}],
    }
    with open(os.path.join(ROOT, "tests", "golden", "kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote tests/golden/kats.json")


if __name__ == "__main__":
    main()
