"""Tools only: run the rest of a measurement script under a KernelPlan of its own (deepsee_amd/plan.py) -- the scripts call
operators directly, without a model that would activate its plan."""
from deepsee_amd import ops

_cm = None


def use_plan(**fields):
    global _cm
    if _cm is not None:
        _cm.__exit__(None, None, None)
    _cm = ops.KernelPlan(**fields).active()
    _cm.__enter__()
