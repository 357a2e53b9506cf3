"""bench.py plumbing on CPU: the op-mix driver runs end to end on the kernel emulator at toy size and
prints one well-formed JSON line.  (No timing claim — this only keeps the GPU-box run from failing on
a Python error.)"""
import json
import os
import sys

import pytest

from conftest import EMU_SO, ROOT


def test_bench_emits_contract_json(built, capsys, monkeypatch):
    sys.path.insert(0, ROOT)
    import zk_dcap_verifier_amd as z
    import bench
    monkeypatch.setattr(z._lib, "LIB_PATH", EMU_SO)
    monkeypatch.setenv("ZK_BENCH_PLUMBING_TEST", "1")
    real_init = z.Backend.__init__

    def small_init(self, device=0, lib_path=None):
        real_init(self, device, lib_path)
        self.tune(msm_sort_threads=32, msm_sort_wgs=2, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
                  msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    monkeypatch.setattr(z.Backend, "__init__", small_init)
    bench.main(["--steps", "1", "--warmup", "0", "--k", "5", "--advice", "3", "--fixed", "2", "--lookups", "1", "--perm-columns", "3",
                "--degree", "4", "--no-extras"])
    out = capsys.readouterr().out.strip().splitlines()[-1]
    line = json.loads(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["unit"] == "proofs/hour" and line["n_gpus"] == 1 and "workload" in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"]
