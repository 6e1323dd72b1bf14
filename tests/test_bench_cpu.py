"""bench.py's host-side choices (no GPU)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_parallelism_follows_the_models_per_rank():
    b = _bench()
    assert b.choose_parallelism(20, 8) == "frames"       # configs[1]: 2.5 models per rank -- replicate, split frames
    assert b.choose_parallelism(200, 8) == "models"      # configs[3]: 25 models per rank
    assert b.choose_parallelism(200, 2) == "models"
    assert b.choose_parallelism(50, 4) == "frames"
    assert b.choose_parallelism(20, 1) == b.choose_parallelism(20, 2)   # the N = 1 line names what N > 1 would run
