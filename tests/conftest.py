import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

# ZK_EMU_LIBDIR: a sanitizer build of the emulator library (tests/run_sanitizers.sh) instead of tests/csrc/libzkmi355_emu.so
EMU_SO = os.path.join(os.environ.get("ZK_EMU_LIBDIR") or os.path.join(ROOT, "tests", "csrc"), "libzkmi355_emu.so")
HOST_SO = os.path.join(ROOT, "tests", "csrc", "libhostharness.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; see oracle/bn254_oracle.c)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pyref():
    import pyref as p
    return p


@pytest.fixture(scope="session")
def built():
    """Make sure the product .so, the emulator .so and the host harness exist (CPU-side build)."""
    import __graft_entry__ as g
    g.build(test_artifacts=True)
    return True


@pytest.fixture(scope="session")
def emu(built):
    """Backend on the kernel EMULATOR (test-only build of the same kernel sources on CPU threads)."""
    import zk_dcap_verifier_amd as z
    be = z.Backend(0, lib_path=EMU_SO)
    assert "EMULATED" in be.version()
    # small launch shapes: every work-item is a pthread here
    be.tune(msm_sort_threads=64, msm_sort_wgs=3, msm_block=32, ntt_threads=32, ntt_tile_log=6, ntt_max_radix_log=4,
            msm_target_threads=64, msm_min_chunk=2, vec_block=32, quot_threads=32)
    yield be
    be.close()


@pytest.fixture(scope="session")
def gpu():
    """Backend on the real GPU through the product library — fails loudly if it is missing."""
    import zk_dcap_verifier_amd as z
    import __graft_entry__ as g
    if not g.library_is_current():              # normally the built library travels with the tree; a binary older than the sources is rebuilt, never tested
        g.build(test_artifacts=False, force=True)
    for exe in ("capi_smoke", "capi_prove"):    # the plain-C consumers link against it
        if not os.path.exists(os.path.join(ROOT, "tests", "csrc", exe)):
            g.build(test_artifacts=True)
            break
    be = z.Backend(0)
    assert "gfx950" in be.version() and "EMULATED" not in be.version()
    if os.environ.get("ZK_TUNE"):                # experiment knob: the whole GPU suite under other tunables, e.g. ZK_TUNE=quot_jit=1 (the generated quotient kernels)
        be.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["ZK_TUNE"].split(",") if kv})
    yield be
    be.close()
