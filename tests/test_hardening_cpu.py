"""CPU-side hardening of the host C++ (SURVEY 5 "race detection / sanitizers": sanitizers on the CPU build only).

* the model loaders (csrc/models.cpp: untrusted .moped.xml text and .mopeddb containers; they replace the reference's
  sXML parser, moped2/libmoped/include/sXML.hpp:66-118, reached from moped2/libmoped/src/moped.cpp:101-137) under
  ASan + UBSan through a mutation fuzz (moped_amd/host/fuzz_models.cpp; `make -C moped_amd/host asan` runs 10 000
  iterations per seed file, here a shorter pass keeps the suite quick);
* the STEP plugin headers compiled against the reference's REAL include/moped.hpp (`make check_ref`), where the
  reference tree exists.

The oracle's sanitizer run is `make -C oracle asan` (the CPU tests under an instrumented liboracle)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "moped_amd", "host")


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_model_loaders_survive_a_mutation_fuzz_under_asan_and_ubsan():
    out = subprocess.run(["make", "-s", "-C", HOST, "asan", "FUZZ_ITERS=1500"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("fuzz_models:")]
    assert len(lines) == 2 and all("no finding" in l for l in lines), out.stdout
    # the fuzz must actually reach the accepting paths (walk / save / reload), not only the error returns
    for l in lines:
        acc = [int(w) for w in l.replace("(", " ").replace(";", " ").split() if w.isdigit()]
        assert acc[0] == 1500 and acc[2] > 100 and acc[4] > 100, l


@pytest.mark.skipif(not os.path.exists("/root/reference/moped2/libmoped/include/moped.hpp"), reason="the reference tree is absent")
def test_plugin_headers_compile_against_the_references_own_moped_hpp():
    out = subprocess.run(["make", "-s", "-B", "-C", HOST, "check_ref"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "compile against the reference's moped.hpp" in out.stdout, (out.stdout, out.stderr[-3000:])
