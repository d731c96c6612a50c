"""Real EuRoC data, when it is there.  Nothing in this repository ships a EuRoC sequence and neither the build container nor
the GPU boxes mount one, so every accuracy figure quoted in README / DESIGN is SYNTHETIC ONLY until this test has run
somewhere: point $EUROC_ROOT at a directory that holds <sequence>/mav0 (e.g. MH_01_easy/mav0) and it runs the headless player on
MH_01_easy with the reference's own configuration and command line, self-initialising, and compares the ATE with the figure the
reference publishes (docs/en/benchmark.md:12: 0.109 m on MH_01; protocol docs/en/tutorials/euroc_evaluation.md:20-41)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EUROC_ROOT = os.environ.get("EUROC_ROOT", "")
SEQ = os.path.join(EUROC_ROOT, "MH_01_easy", "mav0")
PUBLISHED_ATE_MH01 = 0.109

pytestmark = pytest.mark.skipif(not (EUROC_ROOT and os.path.isdir(SEQ)), reason="no EuRoC data ($EUROC_ROOT/MH_01_easy/mav0)")


def _player(lib_dir_player, extra=()):
    cmd = [lib_dir_player, "-sc", os.path.join(ROOT, "configs", "euroc_slam.yaml"), "-dc", os.path.join(ROOT, "configs", "euroc_sensor.yaml"),
           "--tum", os.path.join(ROOT, "gpurun_out", "mh01.tum"), "-p", "euroc://" + SEQ] + list(extra)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
    assert p.returncode == 0, p.stdout + p.stderr
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.gpu
def test_mh01_ate_against_the_published_figure():
    res = _player(os.path.join(ROOT, "xrslam_amd", "bin", "xrslam-player"))
    print("MH_01_easy on MI355X:", res)
    assert res["error"] == "" and res["tracked"] > 0.9 * res["frames"]
    assert 0 <= res["ate_rmse_m"] <= 1.5 * PUBLISHED_ATE_MH01          # north_star: "ATE stays within the reference's reported value"


def test_mh01_cpu_reference_pipeline():
    """The same run through the CPU reference build (host pipeline over the oracle): slow (tens of frames per second)."""
    res = _player(os.path.join(ROOT, "oracle", "_build", "xrslam-player-ref"), extra=["--max-frames", "1200"])
    print("MH_01_easy, CPU reference pipeline, first 1200 frames:", res)
    assert res["error"] == "" and 0 <= res["ate_rmse_m"] <= 0.3
