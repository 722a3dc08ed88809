"""Outlier accounting of the shaded-frame parity tests.

Bit-exactness holds for hit records and Flat frames.  Shaded frames are compared at 1e-4 per channel.  Round 2 still had
three cases with pixels beyond it (39 of 15 360 on the 4K 4-bounce rows, 2 on Cornell depth 5): the bounce samplers'
acosf / sinf / cosf came from the device's libm, which differs from the host's in the last bit on 12-33 % of the arguments, and
a direction that differs in its last bit lets a path leave a silhouette on the other side.  Since round 3 the kernels
evaluate those three with the host libm's own algorithms (rayca_amd/csrc/libm_exact.hpp, bit-identical on every sampler
argument) and EVERY case is at zero (profiles/r03_parity_outliers.json; worst error of any case 3.0e-5, of the path-traced
ones 6.6e-7): tests/parity_bounds.json is empty, and a case without an entry must have no pixel beyond tolerance.  The
counts of the current run are written to gpurun_out/parity_outliers.json."""
import json
import os

import numpy as np

TOL = 1e-4
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_OUT = os.path.join(_ROOT, "gpurun_out", "parity_outliers.json")
_BOUNDS = os.path.join(_ROOT, "tests", "parity_bounds.json")
# RAYCA_PARITY_MEASURE=1: record the counts without asserting (tests/make_parity_bounds.py turns the record into bounds)
MEASURE = os.environ.get("RAYCA_PARITY_MEASURE", "0") == "1"


def _load(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def rel_err(gpu, ora):
    """|gpu - oracle| per channel: absolute inside the displayable range, relative above 1.0."""
    with np.errstate(invalid="ignore"):
        d = np.abs(gpu - ora) / np.maximum(1.0, np.abs(ora))
    both_nan = np.isnan(gpu) & np.isnan(ora)
    return np.where(both_nan, 0.0, np.where(np.isnan(d), np.inf, d))


def check_outliers(name, gpu, ora, tol=TOL):
    """Counts the pixels with any channel beyond `tol`, records the count and compares it with the bound measured for
    `name`.  A test without a recorded bound must be exact to tolerance (bound 0)."""
    d = rel_err(gpu, ora).max(-1)
    n = int((d > tol).sum())
    total = int(d.size)
    worst = float(d.max()) if total else 0.0
    # the same count with the tolerance taken absolutely everywhere (north_star's wording), and how bright the frame gets:
    # recorded next to it, so that what the relative reading above 1.0 lets through is a number and not an argument
    with np.errstate(invalid="ignore"):
        a = np.abs(gpu - ora)
    a = np.where(np.isnan(gpu) & np.isnan(ora), 0.0, np.where(np.isnan(a), np.inf, a)).max(-1)
    n_abs = int((a > tol).sum())
    finite = ora[np.isfinite(ora)]
    report = _load(_OUT)
    report[name] = {"pixels_beyond_tolerance": n, "pixels": total, "fraction": n / max(total, 1), "worst": worst, "tolerance": tol,
                    "pixels_beyond_absolute_tolerance": n_abs, "brightest_oracle_value": float(finite.max()) if finite.size else 0.0}
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    with open(_OUT, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    bound = int(_load(_BOUNDS).get(name, {}).get("bound", 0))
    print(f"[parity] {name}: {n} of {total} pixels beyond {tol:g} (bound {bound}), worst {worst:.3e}; read absolutely: {n_abs}")
    if not MEASURE:
        assert n <= bound, f"{name}: {n} of {total} pixels beyond {tol:g}; the measured bound is {bound} (worst {worst:.3e})"
    return n


def check_u8_outliers(name, u8, ou8):
    """The same accounting after quantisation: pixels with any channel more than 1 LSB apart."""
    d = np.abs(u8.astype(np.int32) - ou8.astype(np.int32)).max(-1)
    n, total = int((d > 1).sum()), int(d.size)
    report = _load(_OUT)
    report[name] = {"pixels_beyond_tolerance": n, "pixels": total, "fraction": n / max(total, 1), "worst": int(d.max()) if total else 0,
                    "tolerance": "1 LSB of RGBA8"}
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    with open(_OUT, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    bound = int(_load(_BOUNDS).get(name, {}).get("bound", 0))
    print(f"[parity] {name}: {n} of {total} pixels more than 1 LSB apart (bound {bound})")
    if not MEASURE:
        assert n <= bound, f"{name}: {n} of {total} pixels more than 1 LSB apart; the measured bound is {bound}"
    return n
