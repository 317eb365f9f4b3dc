"""A fixed slice of the differential fuzzers (tests/tools/fuzz_*.py) inside the GPU suite: the same random cases on every
run -- environment surface, reward nets, wrappers stepping on the environment, hand-coded wrappers, PredictionBonus /
SurpriseBonus with predictor updates (each against the oracle), paired against sequential training steps and StepGraph
replay against the eager loop (the HIP path against itself).  The open-ended runs stay scripts (tests/tools/); their findings become
crafted cases in the other files.  The seeds differ from the tools' default (seed 0), so a plain tool run covers other
cases than this file."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(f"_fuzz_tool_{name}", os.path.join(ROOT, "tests", "tools", f"fuzz_{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("name,cases", [("env", 120), ("nets", 60), ("stack", 40), ("wrappers", 80),
                                        ("train_paths", 25), ("graph", 120), ("prediction", 12)])
def test_fixed_slice_of_the_fuzzer(name, cases):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    assert _tool(name).main(seed0=20260401, cases=cases) == cases
