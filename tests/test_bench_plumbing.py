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
    bench.main(["--mode", "opmix", "--steps", "1", "--warmup", "0", "--k", "5", "--advice", "3", "--fixed", "2", "--lookups", "1", "--perm-columns", "3",
                "--degree", "4", "--no-extras"])
    out = capsys.readouterr().out.strip().splitlines()[-1]
    line = json.loads(out)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["unit"] == "proofs/hour" and line["n_gpus"] == 1 and "workload" in line["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"]


def test_bench_prove_mode_emits_verified_proof(built, capsys, monkeypatch):
    """The default mode (real create_proof) end to end on the emulator, with the circuit generator shrunk to 2 + 2 advice
    columns at k = 6: the JSON line carries the contract keys and the CPU leg's verify_proof accepted the proof."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import zk_dcap_verifier_amd as z
    import bench
    import sgx_shaped_circuit as sgx
    monkeypatch.setattr(z._lib, "LIB_PATH", EMU_SO)
    monkeypatch.setenv("ZK_BENCH_PLUMBING_TEST", "1")
    for name, val in (("N_GATE_COLS", 2), ("N_LOOKUP_COLS", 2), ("N_FIXED", 7), ("N_GATES", 3)):
        monkeypatch.setattr(sgx, name, val)
    real_init = z.Backend.__init__

    def small_init(self, device=0, lib_path=None):
        real_init(self, device, lib_path)
        self.tune(msm_sort_threads=32, msm_sort_wgs=2, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
                  msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    monkeypatch.setattr(z.Backend, "__init__", small_init)
    monkeypatch.setattr(bench, "msm_microbench", lambda *a, **k: {"skipped": "plumbing test"})
    monkeypatch.setattr(bench, "PLUMBING_SKIP_NTT22", True, raising=False)
    bench.main(["--steps", "1", "--warmup", "0", "--k", "6", "--inflight", "1"])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line["config"]["mode"] == "prove" and line["unit"] == "proofs/hour"
    assert line["cpu_baseline"]["verify_proof"]["accepted"] is True
    assert line["roofline"]["kernel"] == "msm_accumulate_kernel"
    # the PCIe-inclusive regions (witness starting in host memory) and the thin-shim replay (per-call host-buffer entry points + CPU port of a13-a16)
    assert set(line["extra"]["host_witness"]) >= {"pinned", "pageable", "bytes_per_proof"}
    c1 = line["extra"]["cfg1_p256_k18"]                                                                # BASELINE configs[0]'s shape: Poseidon transcript, h(X) from three cosets, checked by the oracle's verifier
    assert c1["verify_proof_accepted"] is True and c1["pieces_from_cosets"] is True and c1["proof_bytes"] == 1504, c1
    ts = line["extra"]["thin_shim"]
    assert ts["calls"] == {"zk_msm": line["extra"]["ops_per_proof"]["msm"], "zk_ntt": line["extra"]["ops_per_proof"]["intt_2^k"], "zk_evaluate_h": 1}
    assert ts["proofs_per_hour"] > 0 and line["cpu_baseline"]["a13_a16_cpu_port"]["total_ms"] > 0
    # the rung between the thin shim and `value`: phase-batched device calls on host-resident columns (real transfers, same proof bytes), with and without the shim's device cache
    pb = line["extra"]["phase_batched_shim"]
    for rung in ("host_columns", "host_columns_with_device_cache"):
        assert pb[rung]["same_bytes_as_the_resident_prover"] is True and pb[rung]["proofs_per_hour"] > 0 and pb[rung]["pcie_bytes_down"] > 0, pb[rung]
    assert pb["host_columns"]["pcie_bytes_up"] > pb["host_columns_with_device_cache"]["pcie_bytes_up"]


def _bench_rank(rank, world, port, out_dir):
    import io
    import contextlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      ZK_BENCH_PLUMBING_TEST="1", ZK_BENCH_SHARDED_LOG_N="7")
    sys.path.insert(0, ROOT)
    import zk_dcap_verifier_amd as z
    import bench
    z._lib.LIB_PATH = EMU_SO
    real_init = z.Backend.__init__

    def small_init(self, device=0, lib_path=None):
        real_init(self, device, lib_path)
        self.tune(msm_sort_threads=32, msm_sort_wgs=2, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
                  msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    z.Backend.__init__ = small_init
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--mode", "opmix", "--gpus", str(world), "--steps", "1", "--warmup", "0", "--k", "4", "--advice", "2", "--fixed", "2", "--lookups", "1",
                    "--perm-columns", "2", "--degree", "4"])
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(buf.getvalue())


def test_bench_two_ranks_gloo(built, tmp_path):
    """The N > 1 launch path of bench.py (barrier, max-over-ranks timing, sharded MSM + all_gather) with two
    gloo ranks on the emulator; rank 0 alone prints the JSON line."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bench_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out0 = open(tmp_path / "rank0.txt").read().strip().splitlines()
    line = json.loads(out0[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["extra"]["msm_sharded_2^7"]["closed_form_check"] is True, line["extra"]
    assert open(tmp_path / "rank1.txt").read().strip() == ""


def _bench_rank_prove(rank, world, port, out_dir):
    import io
    import contextlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      ZK_BENCH_PLUMBING_TEST="1", ZK_BENCH_SHARDED_LOG_N="7")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import zk_dcap_verifier_amd as z
    import bench
    import sgx_shaped_circuit as sgx
    z._lib.LIB_PATH = EMU_SO
    sgx.N_GATE_COLS, sgx.N_LOOKUP_COLS, sgx.N_FIXED, sgx.N_GATES = 2, 2, 7, 3
    real_init = z.Backend.__init__

    def small_init(self, device=0, lib_path=None):
        real_init(self, device, lib_path)
        self.tune(msm_sort_threads=32, msm_sort_wgs=2, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
                  msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    z.Backend.__init__ = small_init
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.main(["--gpus", str(world), "--steps", "1", "--warmup", "0", "--k", "6", "--inflight", "1"])
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(buf.getvalue())


def test_bench_prove_mode_two_ranks_sharded_proof(built, tmp_path):
    """N = 2 in the default mode: replicas prove their own streams, then ONE proof is spread over both ranks (sharded SRS tables,
    all-gathered partial commitments) and must be byte-identical to the single-device proof of the same RNG stream."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_bench_rank_prove, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    line = json.loads([l for l in open(tmp_path / "rank0.txt").read().strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["mode"] == "prove"
    assert line["extra"]["sharded_proof"]["identical_to_single_gpu_proof"] is True, line["extra"]
    assert line["extra"]["sharded_proof"]["prover"].startswith("native"), line["extra"]           # zk_plonk_create_proof in shard mode, collective through the callback
    assert line["extra"]["sharded_proof"]["allgathers_per_proof"] == 7 + 1, line["extra"]          # one per commitment phase + the quotient's numerators
    assert line["extra"]["msm_sharded_2^7"]["closed_form_check"] is True
