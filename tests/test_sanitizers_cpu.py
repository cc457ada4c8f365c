"""CPU sanitizer run: the oracle restatements and the product's HOST-side code
that takes untrusted text or sizes - the params.yaml reader behind slam_config_load, filter_driver's message-stream reader,
the scenario generators - built with AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`) and run over
well-formed and malformed inputs.  A sanitizer report aborts the binary (-fno-sanitize-recover=all)."""
import os
import subprocess

from conftest import ROOT


def test_oracle_and_host_parsers_are_clean_under_asan_ubsan(tmp_path):
    mk = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert mk.returncode == 0, mk.stdout[-2000:] + mk.stderr[-2000:]
    exe = os.path.join(ROOT, "oracle", "_asan", "oracle_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([exe, str(tmp_path), os.path.join(ROOT, "live_ekf_slam_amd", "data", "fixed_maps.json")],
                         capture_output=True, text=True, timeout=900, env=env)
    text = out.stdout + out.stderr
    assert "ERROR: AddressSanitizer" not in text and "runtime error:" not in text and "LeakSanitizer" not in text, text[-3000:]
    assert out.returncode == 0, text[-3000:]
    assert "0 failed" in out.stdout
