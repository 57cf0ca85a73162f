"""Host-side helpers of bench.py that need no GPU."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location(
        "bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_numa_pinning_is_a_no_op_without_a_gpu_and_never_raises():
    """pin_to_gpu_numa_node: on a box whose GPU topology cannot be read (this
    container has no GPU) it returns None and leaves the affinity mask alone;
    where it can, the new mask is a non-empty subset of the old one."""
    bench = _bench()
    before = os.sched_getaffinity(0)
    try:
        got = bench.pin_to_gpu_numa_node(0)
        after = os.sched_getaffinity(0)
        if got is None:
            assert after == before
        else:
            assert after and after <= before
    finally:
        os.sched_setaffinity(0, before)


def test_bench_defaults_are_the_documented_driver_configuration():
    """The driver runs `bench.py --gpus N --steps K --warmup W` and nothing
    else: the defaults ARE the measured configuration."""
    bench = _bench()
    import sys
    argv = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
    finally:
        sys.argv = argv
    assert a.gpus == 1
    assert a.batch == 8000 and a.frames_per_launch == 12
    assert a.block_count == 50000  # the reference's advised map size
    assert a.dist_backend == "nccl"


def test_driver_line_stays_under_6_kb():
    """The driver keeps ~8 KB of the line: headline + roofline + cpu_baseline +
    one compact object per BASELINE config must fit in 6 KB, small objects
    first; per-kernel tables and notes live in bench_detail.json. Replayed on
    last round's full record (profiles/r3z_bench.json, 16 KB)."""
    import json
    bench = _bench()
    with open(os.path.join(ROOT, "profiles", "r3z_bench.json")) as f:
        d = json.load(f)
    sec = d.pop("secondary")
    d["roofline"].pop("note", None)
    line = bench.compact_line(d, sec)
    text = json.dumps(line)
    assert len(text) < 6144, len(text)
    keys = list(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
              "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    # the per-config objects come before the long ones
    assert keys.index("configs0") < keys.index("config") < \
        keys.index("roofline")
    assert line["configs2"]["1280x720"]["frames_per_s"] > 0
    assert line["configs2"]["640x480"]["gpu_busy_frac"] > 0
    assert line["configs4"]["frames_per_s"] > 0
    assert "per_kernel" not in text and "note" not in line["roofline"]
