"""Runs the GPU test modules whose kernels have never been on a device (tests/conftest.py: ISOLATED_GPU_MODULES) in a child
process with a timeout: a memory fault or a hang there fails THIS test and leaves the session -- and the results of every
validated test before it -- intact.  The file sorts last for the same reason."""
import os
import signal
import subprocess
import sys

import pytest

from conftest import ISOLATED_GPU_MODULES, ROOT

pytestmark = pytest.mark.gpu


def test_unvalidated_gpu_modules_in_a_child_process():
    if not ISOLATED_GPU_MODULES:
        pytest.skip("no GPU module is waiting for its first device run")
    files = [os.path.join(ROOT, "tests", m) for m in ISOLATED_GPU_MODULES]
    env = dict(os.environ, GORSE_GPU_ISOLATED="1")
    child = subprocess.Popen([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + files, cwd=ROOT, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=240)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)
        out, _ = child.communicate()
        pytest.fail("the isolated GPU modules did not finish within 240 s (killed)\n" + (out or "")[-4000:])
    print(out[-6000:])  # shown with -rP / on failure: the child's own summary
    assert child.returncode == 0, "isolated GPU modules: exit code %d\n%s" % (child.returncode, out[-6000:])
