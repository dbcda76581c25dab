"""The tracing kernels compute 1/x, sqrt(x), x/pi, byte/255 and the path-index divisions with short forms (hardware
approximation plus one residual correction; a multiplier for the indices) instead of the IEEE expansions.  Results
must keep the bits of the correctly rounded operations, which is what the oracle computes on the CPU: the library
checks that on the GPU it runs on, over every binary32 pattern."""
import pytest

pytestmark = pytest.mark.gpu


def test_short_arithmetic_forms_are_exact_over_all_inputs():
    from pt_amd import hip
    ctx = hip.Context(0)
    try:
        bad = ctx.selftest_arith()
    finally:
        ctx.close()
    assert bad == {"reciprocal": 0, "sqrt": 0, "div_pi": 0, "unorm8": 0, "index_division": 0}
