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
    assert a.block_count == 524288
    assert a.dist_backend == "nccl"
