"""Instance-scoped entry points (XRSLAMAmdInstance*, include/XRSLAM.h): several independent sequences in one process.

The reference is a process singleton (xrslam-interface/src/XRSLAMManager.cpp:6-9) and keeps solver configuration,
CLAHE / GFTT objects, id counters and RD-VIO statistics in statics; here that state lives in the instance.
The tests drive two instances from two threads, interleaved with each other, and require every sequence to come out
exactly as it does alone through the reference's six global symbols.  CPU: the host pipeline over the oracle;
GPU: the product library (kernels of the two instances overlap on the device)."""
import os
import threading

import numpy as np
import pytest

from xrslam_amd.harness import runner, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libxrslam_oracle.so")
N = 56


def _drain(s):
    while s.step():
        assert not s.error(), s.error()
    s.flush()
    assert not s.error(), s.error()
    t = s.times()
    out = np.array(s.poses), (t.frames, t.solves, t.solve_iterations, t.marginalizations, t.keyframes)
    s.close()
    return out


def _alone_and_together(lib_path):
    seqs = [scene.make_sequence(n_frames=N, seed=sd) for sd in (1, 2)]
    alone = [_drain(runner.Session(lib_path, q)) for q in seqs]           # the reference's global instance, one after the other
    sessions = [runner.Session(lib_path, q, instance=True) for q in seqs]   # two live instances at once
    res, errs = [None, None], []

    def work(i):
        try:
            res[i] = _drain(sessions[i])
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    return alone, res


def _check(alone, together):
    for (pa, ca), (pt, ct) in zip(alone, together):
        assert ca == ct                                  # identical discrete decisions
        assert pa.shape == pt.shape and len(pa) >= N - 40
        np.testing.assert_array_equal(pa, pt)            # same arithmetic, same order: bit-identical poses
    assert not np.array_equal(together[0][0][:, 1:4], together[1][0][:, 1:4])   # and they are different sequences


def test_two_instances_in_one_process_cpu():
    if not os.path.exists(ORACLE_LIB):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    _check(*_alone_and_together(ORACLE_LIB))


@pytest.mark.gpu
def test_two_instances_in_one_process_gpu():
    from xrslam_amd import _lib
    _check(*_alone_and_together(_lib.LIB_PATH))


def test_instance_create_reports_errors():
    lib = runner.load(ORACLE_LIB)
    import ctypes as C
    h, cfg = C.c_void_p(), C.c_void_p()
    assert lib.XRSLAMAmdInstanceCreate(b"/nonexistent/slam.yaml", b"/nonexistent/sensor.yaml", C.byref(h), C.byref(cfg)) == 0
    assert not h.value
    assert lib.XRSLAMAmdLastError().decode() != ""
