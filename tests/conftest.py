import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-to-action-release_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------- parity ledger
# Every measured error a parity test computes (its `rel()` / `close()` helper, or an explicit `parity_record`) is written, per test, to
# gpurun_out/r06_parity.json at the end of the session (copied to profiles/ per round): `pytest -q` prints nothing on success, the ledger
# shows the margins -- and which bound carried a test that accepts the larger of two.
_LEDGER = {}
_CURRENT = [None]


@pytest.fixture(autouse=True)
def _parity_ledger_context(request):
    _CURRENT[0] = request.node.nodeid
    yield
    _CURRENT[0] = None


def parity_record(what, value, bound=None, **info):
    """Note a measured error of the running test (value: float; bound: the tolerance it is compared with, if known)."""
    if _CURRENT[0] is None:
        return value
    ent = {"what": str(what)[:80], "value": float(value)}
    if bound is not None:
        ent["bound"] = float(bound)
    ent.update({k: (float(v) if isinstance(v, (int, float)) else str(v)) for k, v in info.items()})
    rows = _LEDGER.setdefault(_CURRENT[0], [])
    if len(rows) < 64:
        rows.append(ent)
    else:                                           # long loops: keep the worst
        worst = min(range(len(rows)), key=lambda i: rows[i]["value"])
        if rows[worst]["value"] < ent["value"]:
            rows[worst] = ent
    return value


def pytest_sessionfinish(session, exitstatus):
    if not _LEDGER:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, "r06_parity.json")
        old = {}
        if os.path.exists(path):                    # several pytest invocations of one GPU call accumulate
            try:
                old = json.load(open(path)).get("tests", {})
            except Exception:
                old = {}
        old.update(_LEDGER)
        summary = {k: {"max": max(r["value"] for r in v), "n": len(v)} for k, v in old.items()}
        with open(path, "w") as f:
            json.dump({"unit": "as computed by the test (mostly max |a - b| / max |b|); the helpers are also used for 'must differ' checks (a changed "
                               "weight must change the output: values >> 1e-2 are those), so read a value next to its test's assertion",
                       "summary": summary, "tests": old}, f, indent=0, sort_keys=True)
    except Exception as e:                          # never fail a session over the ledger
        print(f"[parity ledger] not written: {e}")
