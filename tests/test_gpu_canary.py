"""Memory-safety harness of the GPU suite ("race detection / sanitizers": GPU AddressSanitizer is not available on this
pool, so the suite brings its own).  tests/canary/canary_alloc.cpp is a guard-zone device allocator: under VIDI_CANARY=1 every tensor of the
session is its own hipMalloc with poisoned zones before and behind it, compared with the pattern after every test (tests/conftest.py).

* `test_canary_negative_control_*` (runs only under VIDI_CANARY): a C-ABI call that is told a row count 8 elements longer than its output
  tensor must be caught — the harness sees a 16-byte stray store;
* `test_kernel_and_model_suites_under_the_guard_zone_allocator` (the ordinary `-m gpu` run): re-runs the per-kernel, model, Vidi-7B and
  real-dims parity files in a child pytest with VIDI_CANARY=1 — every entry point of include/vidi_hip.h, both dtypes, ragged / tiny / real
  shapes — and fails if any tensor's zones were touched.  `tools/gpu_round.sh canary` runs the WHOLE suite that way (profiles/r6_canary.log)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CANARY = os.environ.get("VIDI_CANARY", "0") in ("1", "2")


@pytest.mark.skipif(not CANARY, reason="needs the guard-zone allocator (VIDI_CANARY=1)")
@pytest.mark.parametrize("extra", [8, 4096])
def test_canary_negative_control_catches_a_stray_store(extra):
    import canary
    from vidi_amd import hip
    lib = hip.load_library()
    n = 1000
    x = torch.ones(n + extra, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
    assert canary.check()[0] == 0
    # vidi_scale(x, out, count, ...): the count says n + extra, the output tensor holds n elements
    rc = lib.vidi_scale(hip._p(x), hip._p(out), n + extra, 2.0, hip.DT_BF16, hip._stream())
    assert rc == 0
    torch.cuda.synchronize()
    nviol, msg, live, total = canary.check()
    assert nviol >= 1 and "PAST the end" in msg, (nviol, msg)
    canary.reset()
    assert canary.check()[0] == 0                                   # re-armed: reported once
    assert float(out.float().sum()) == 2.0 * n


@pytest.mark.skipif(not CANARY, reason="needs the guard-zone allocator (VIDI_CANARY=1)")
def test_canary_allocator_is_the_one_serving_tensors():
    import canary
    before = canary.check()[3]
    t = torch.empty(123457, dtype=torch.float32, device="cuda")
    assert canary.check()[3] == before + 1 and t.data_ptr() % 256 == 0
    del t


@pytest.mark.skipif(CANARY, reason="this IS the child run")
def test_kernel_and_model_suites_under_the_guard_zone_allocator():
    files = ["test_gpu_kernels.py", "test_gpu_model.py", "test_gpu_vidi7b.py", "test_realdims_golden.py", "test_gpu_canary.py"]
    env = dict(os.environ, VIDI_CANARY="1")
    env.pop("VIDI_TEST_REPORT", None)                                 # the audit log belongs to the parent run
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"] + [os.path.join(HERE, f) for f in files],
                       capture_output=True, text=True, timeout=3000, env=env, cwd=os.path.dirname(HERE))
    tail = (r.stdout or "")[-3000:]
    log = os.environ.get("VIDI_CANARY_LOG")
    if log:
        with open(log, "w") as f:
            f.write(r.stdout or "")
    assert r.returncode == 0, tail + (r.stderr or "")[-2000:]
    assert " passed" in tail and "canary:" not in tail, tail
